// 3x3 convolution kernels for the IODINE decoder / refinement stacks on gfx950.
//
// Replaces the reference's nn.Conv2d call sites (lib/modeling/iodine.py:422,435,583,592 and
// their autograd backward) with fp32 MFMA implicit-GEMM kernels over NHWC activations:
//   M = output pixels, N = output channels, K = 9 taps x input channels.
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain, so results differ from the CPU path only
// by summation order.
#include "common.h"
#include "pack_bodies.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <type_traits>

// =========================================================================================
// weight packer: OIHW (or its dgrad transform) -> wpk[chunk][quad][co] float4
// =========================================================================================
__global__ void pack_conv_weights_kernel(const float* __restrict__ src, int O, int I, int cin_pad,
                                         int cout, int tflip, float* __restrict__ dst)
{
    const int cc = conv_cc(cin_pad), qpt = cc / 4, nq = 9 * qpt, nqp = (nq + 1) & ~1;
    const size_t total = (size_t)(cin_pad / cc) * nqp * cout * 4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        size_t r = idx >> 2;
        const int co = r % cout; r /= cout;
        const int q = r % nqp;
        const int chunk = r / nqp;
        const int tap = q / qpt, cig = q % qpt;
        const int ci = chunk * cc + cig * 4 + e;
        float v = 0.f;
        if (q < nq) {
            if (!tflip) {                       // forward: conv-in = I axis, conv-out = O axis
                if (ci < I && co < O) v = src[((size_t)co * I + ci) * 9 + tap];
            } else if (tflip == 1) {            // stride-1 dgrad: conv-in = O axis, conv-out = I axis, taps flipped
                if (ci < O && co < I) v = src[((size_t)ci * I + co) * 9 + (8 - tap)];
            } else {                            // strided (transposed-conv) dgrad: axes swapped, taps as they are
                if (ci < O && co < I) v = src[((size_t)ci * I + co) * 9 + tap];
            }
        }
        dst[idx] = v;
    }
}

hipError_t launch_pack_conv_weights(hipStream_t st, const float* src, int O, int I, int cin_pad, int cout,
                                    int tflip, float* dst)
{
    const size_t total = conv_wpk_elems(cin_pad, cout) * 4;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_conv_weights_kernel, dim3(blocks), dim3(256), 0, st, src, O, I, cin_pad, cout,
                       tflip, dst);
    return hipGetLastError();
}

// =========================================================================================
// LDS-tiled stride-1 3x3 conv, pad 1 (decoder layers, and their dgrad with transformed weights)
//   block = 256 threads (4 waves), output tile 16x16 pixels x COUT channels
//   wave w owns tile rows 4w..4w+3: two 32-pixel MFMA row-blocks (2 rows x 16 cols each)
//   per channel chunk: stage (18x18 halo) x CC input channels + the chunk's packed weights in LDS
// =========================================================================================
template <int CIN, int COUT, int EPI>
__global__ __launch_bounds__(256, 2)
void conv3x3_tile_kernel(const float* __restrict__ in, const float4* __restrict__ wpk,
                         const float* __restrict__ bias, const float* __restrict__ aux,
                         float* __restrict__ out, int S, int tiles)
{
    constexpr int CC = conv_cc(CIN);
    constexpr int NCHUNK = CIN / CC;
    constexpr int QPT = CC / 4;
    constexpr int NQ = 9 * QPT;
    constexpr int NQP = (NQ + 1) & ~1;
    constexpr int NPAIR = NQP / 2;
    constexpr int CP = (CC == 4) ? 4 : CC + 4;       // LDS pixel stride in floats (bank spread)
    constexpr int NT = COUT / 32;
    constexpr int HALO = 18;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float4* s_w = reinterpret_cast<float4*>(smem + HALO * HALO * CP);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, li = lane & 31;
    const int prow = li >> 4, pcol = li & 15;

    int bid = blockIdx.x;
    const int tx = bid % tiles; bid /= tiles;
    const int ty = bid % tiles;
    const int n = bid / tiles;
    const int y0 = ty * 16 - 1, x0 = tx * 16 - 1;
    const float* in_n = in + (size_t)n * S * S * CIN;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        if (chunk > 0) __syncthreads();
        // ---- stage the input halo tile for this channel chunk ----
        for (int idx = tid; idx < HALO * HALO * QPT; idx += 256) {
            const int px = idx / QPT, cq = idx % QPT;
            const int hy = px / HALO, hx = px % HALO;
            const int gy = y0 + hy, gx = x0 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < S && gx >= 0 && gx < S)
                v = *reinterpret_cast<const float4*>(in_n + ((size_t)gy * S + gx) * CIN + chunk * CC + cq * 4);
            *reinterpret_cast<float4*>(s_in + px * CP + cq * 4) = v;
        }
        // ---- stage this chunk's packed weights ----
        const float4* wsrc = wpk + (size_t)chunk * NQP * COUT;
        for (int idx = tid; idx < NQP * COUT; idx += 256) s_w[idx] = wsrc[idx];
        __syncthreads();

#pragma unroll
        for (int g = 0; g < NPAIR; ++g) {
            const int q = 2 * g + half;
            const int qa = q < NQ ? q : NQ - 1;          // padded quad: weights are zero, read any valid pixel
            const int tap = qa / QPT, cig = qa % QPT;
            const int dy = tap / 3, dx = tap % 3;
            float4 a[2], b[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int hy = 4 * wv + 2 * mt + prow + dy, hx = pcol + dx;
                a[mt] = *reinterpret_cast<const float4*>(s_in + (hy * HALO + hx) * CP + cig * 4);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = s_w[q * COUT + nt * 32 + li];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, b[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, b[nt].w, acc[mt][nt], 0, 0, 0);
                }
        }
    }

    // ---- epilogue: D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) ----
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 32 + li;
            float bv = 0.f;
            if (EPI == EPI_BIAS_ELU) bv = bias[co];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;        // 0..31 inside the row-block
                const int gy = ty * 16 + 4 * wv + 2 * mt + (m >> 4);
                const int gx = tx * 16 + (m & 15);
                const size_t o = (((size_t)n * S + gy) * S + gx) * COUT + co;
                float v = acc[mt][nt][r];
                if (EPI == EPI_BIAS_ELU) v = elu1(v + bv);
                else if (EPI == EPI_MUL_ELUGRAD) v *= elu1_grad_from_out(aux[o]);
                out[o] = v;
            }
        }
}

template <int CIN, int COUT, int EPI>
static hipError_t launch_tile_inst(hipStream_t st, const float* in, const float* wpk, const float* bias,
                                   const float* aux, float* out, int N, int S)
{
    constexpr int CC = conv_cc(CIN);
    constexpr int CP = (CC == 4) ? 4 : CC + 4;
    constexpr size_t lds = (size_t)(18 * 18 * CP) * 4 + (size_t)conv_nqp(CIN) * COUT * 16;
    static std::atomic<unsigned> attr_devs{0};                             // devices this instance is configured on
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_tile_kernel<CIN, COUT, EPI>, (int)lds, attr_devs); e != hipSuccess) return e;
    const int tiles = S / 16;
    hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, COUT, EPI>), dim3(N * tiles * tiles), dim3(256), lds, st, in,
                       reinterpret_cast<const float4*>(wpk), bias, aux, out, S, tiles);
    return hipGetLastError();
}

hipError_t launch_conv3x3_tile(hipStream_t st, const float* in, const float* wpk, const float* bias,
                               const float* aux, float* out, int N, int S, int cin, int cout, int epi)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
#define TILE_CASE(CI, CO, EP) \
    if (cin == CI && cout == CO && epi == EP) return launch_tile_inst<CI, CO, EP>(st, in, wpk, bias, aux, out, N, S);
    TILE_CASE(64, 64, EPI_BIAS_ELU) TILE_CASE(64, 64, EPI_MUL_ELUGRAD)
    TILE_CASE(32, 32, EPI_BIAS_ELU) TILE_CASE(32, 32, EPI_MUL_ELUGRAD)
    TILE_CASE(4, 64, EPI_MUL_ELUGRAD) TILE_CASE(4, 32, EPI_MUL_ELUGRAD)
#undef TILE_CASE
    return hipErrorInvalidValue;
}

// =========================================================================================
// gather-style 3x3 conv with stride (refinement network, lib/modeling/iodine.py:459,480,583):
//   M = linear output pixel index over the whole slot batch, A operand gathered straight from
//   global memory (L1/L2 absorb the 9/stride^2 re-reads), weights chunk-staged in LDS.
//   block = 256 threads, 4 waves x 32 output pixels; epilogue bias + ELU.
// =========================================================================================
// MODE 0: forward conv (in = layer input, out = layer output, epilogue bias + ELU).
// MODE 1: data gradient of the same strided conv (in = d(pre-activation) of the layer, (IH, IW) its size;
//         out = gradient wrt the layer input of size (OH, OW), epilogue multiplies by ELU'(aux)):
//         dIn[q] = sum_tap d[(q + 1 - tap) / STRIDE] . W[:, :, tap] over the taps where the division is exact.
template <int CIN, int COUT, int STRIDE, int MODE>
__global__ __launch_bounds__(256, 2)
void conv3x3_gather_kernel(const float* __restrict__ in, const float4* __restrict__ wpk,
                           const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                           int M, int IH, int IW, int OH, int OW)
{
    constexpr int CC = conv_cc(CIN);
    constexpr int NCHUNK = CIN / CC;
    constexpr int QPT = CC / 4;
    constexpr int NQ = 9 * QPT;
    constexpr int NQP = (NQ + 1) & ~1;
    constexpr int NPAIR = NQP / 2;
    constexpr int NT = COUT / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* s_w = reinterpret_cast<float4*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, li = lane & 31;
    // MODE 0: blocks walk the output pixels linearly.  MODE 1 (STRIDE 2, even OH/OW): blocks are grouped by the
    // parity class (oy & 1, ox & 1) of the pixels they own, because a pixel only receives the taps with
    // tap_y = oy + 1 (mod 2), tap_x = ox + 1 (mod 2): 1 / 2 / 2 / 4 of the 9 taps -> 4x fewer MFMAs, wave-uniform.
    const int OH2 = OH / 2, OW2 = OW / 2, Mc = M / 4;
    const int bpc = (Mc + 127) / 128;
    const int cls = MODE == 1 ? blockIdx.x / bpc : 0;
    const int cy = cls >> 1, cx = cls & 1;
    const int mbase = MODE == 1 ? (blockIdx.x % bpc) * 128 : blockIdx.x * 128;
    const int Mlim = MODE == 1 ? Mc : M;
    auto decode = [&](int mloc, int& n_, int& oy_, int& ox_) {
        if (MODE == 1) {
            ox_ = 2 * (mloc % OW2) + cx; mloc /= OW2;
            oy_ = 2 * (mloc % OH2) + cy; n_ = mloc / OH2;
        } else {
            ox_ = mloc % OW; mloc /= OW;
            oy_ = mloc % OH; n_ = mloc / OH;
        }
    };
    const int m = mbase + wv * 32 + li;
    const bool mvalid = m < Mlim;
    int n, oy, ox;
    decode(mvalid ? m : 0, n, oy, ox);
    const float* in_n = in + (size_t)n * IH * IW * CIN;
    const int iy0 = MODE == 0 ? oy * STRIDE - 1 : oy + 1, ix0 = MODE == 0 ? ox * STRIDE - 1 : ox + 1;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        if (chunk > 0) __syncthreads();
        const float4* wsrc = wpk + (size_t)chunk * NQP * COUT;
        for (int idx = tid; idx < NQP * COUT; idx += 256) s_w[idx] = wsrc[idx];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < NPAIR; ++g) {
            if (MODE == 1 && (QPT % 2) == 0) {           // both quads of the pair belong to tap g / (QPT/2)
                const int tapg = (2 * g) / QPT;
                if (((cy + 1 - tapg / 3) & 1) || ((cx + 1 - tapg % 3) & 1)) continue;      // block-uniform
            }
            const int q = 2 * g + half;
            const int qa = q < NQ ? q : NQ - 1;
            const int tap = qa / QPT, cig = qa % QPT;
            int iy, ix;
            bool ok = mvalid && q < NQ;
            if (MODE == 0) {
                iy = iy0 + tap / 3; ix = ix0 + tap % 3;
            } else {
                const int ty2 = iy0 - tap / 3, tx2 = ix0 - tap % 3;          // = STRIDE * (source index) when exact
                ok = ok && ty2 >= 0 && tx2 >= 0 && (ty2 % STRIDE) == 0 && (tx2 % STRIDE) == 0;
                iy = ty2 / STRIDE; ix = tx2 / STRIDE;
            }
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && iy >= 0 && iy < IH && ix >= 0 && ix < IW)
                a = *reinterpret_cast<const float4*>(in_n + ((size_t)iy * IW + ix) * CIN + chunk * CC + cig * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 b = s_w[q * COUT + nt * 32 + li];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nt], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 32 + li;
        const float bv = MODE == 0 ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = mbase + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (mm < Mlim) {
                size_t pix = (size_t)mm;
                if (MODE == 1) {
                    int n2, oy2, ox2;
                    decode(mm, n2, oy2, ox2);
                    pix = ((size_t)n2 * OH + oy2) * OW + ox2;
                }
                const size_t o = pix * COUT + co;
                out[o] = MODE == 0 ? elu1(acc[nt][r] + bv) : acc[nt][r] * elu1_grad_from_out(aux[o]);
            }
        }
    }
}

template <int CIN, int COUT, int STRIDE>
static hipError_t launch_gather_inst(hipStream_t st, const float* in, const float* wpk, const float* bias,
                                     float* out, int N, int IH, int IW)
{
    constexpr size_t lds = (size_t)conv_nqp(CIN) * COUT * 16;
    const int OH = (IH + 2 - 3) / STRIDE + 1, OW = (IW + 2 - 3) / STRIDE + 1;
    const int M = N * OH * OW;
    hipLaunchKernelGGL((conv3x3_gather_kernel<CIN, COUT, STRIDE, 0>), dim3((M + 127) / 128), dim3(256), lds, st, in,
                       reinterpret_cast<const float4*>(wpk), bias, nullptr, out, M, IH, IW, OH, OW);
    return hipGetLastError();
}

// data gradient of a strided conv whose INPUT was (big_h, big_w): d has the conv's output size
template <int CIN, int COUT, int STRIDE>
static hipError_t launch_gather_dgrad_inst(hipStream_t st, const float* d, const float* wpk, const float* aux,
                                           float* out, int N, int big_h, int big_w)
{
    constexpr size_t lds = (size_t)conv_nqp(CIN) * COUT * 16;
    const int dh = (big_h - 1) / STRIDE + 1, dw = (big_w - 1) / STRIDE + 1;
    const int M = N * big_h * big_w;
    if ((big_h & 1) || (big_w & 1)) return hipErrorInvalidValue;
    const int bpc = (M / 4 + 127) / 128;
    hipLaunchKernelGGL((conv3x3_gather_kernel<CIN, COUT, STRIDE, 1>), dim3(4 * bpc), dim3(256), lds, st, d,
                       reinterpret_cast<const float4*>(wpk), nullptr, aux, out, M, dh, dw, big_h, big_w);
    return hipGetLastError();
}

hipError_t launch_conv3x3_gather_dgrad(hipStream_t st, const float* d, const float* wpk, const float* aux, float* out,
                                       int N, int big_h, int big_w, int c, int stride)
{
    if (stride != 2) return hipErrorInvalidValue;
    if (c == 64) return launch_gather_dgrad_inst<64, 64, 2>(st, d, wpk, aux, out, N, big_h, big_w);
    if (c == 32) return launch_gather_dgrad_inst<32, 32, 2>(st, d, wpk, aux, out, N, big_h, big_w);
    return hipErrorInvalidValue;
}

hipError_t launch_conv3x3_gather(hipStream_t st, const float* in, const float* wpk, const float* bias,
                                 float* out, int N, int IH, int IW, int cin, int cout, int stride)
{
#define G_CASE(CI, CO, SD) \
    if (cin == CI && cout == CO && stride == SD) return launch_gather_inst<CI, CO, SD>(st, in, wpk, bias, out, N, IH, IW);
    G_CASE(20, 64, 2) G_CASE(64, 64, 2) G_CASE(20, 32, 2) G_CASE(32, 32, 2)
#undef G_CASE
    return hipErrorInvalidValue;
}

// =========================================================================================
// decoder output conv C -> 4 (rgb pre-sigmoid x3, mask logit), lib/modeling/iodine.py:422,435.
// N=4 would waste 7/8 of a 32-wide MFMA tile, so this one is a VALU kernel: one thread per
// output pixel, weights wave-uniform from LDS (broadcast reads), input straight from L1/L2.
//   wk layout: [tap][ci][4]
// =========================================================================================
template <int C>
__global__ __launch_bounds__(256)
void dec_out_kernel(const float* __restrict__ in, const float4* __restrict__ wk, const float* __restrict__ bias,
                    float4* __restrict__ out, int S, int tiles)
{
    __shared__ float4 s_w[9 * C];
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * C; i += 256) s_w[i] = wk[i];
    __syncthreads();
    int bid = blockIdx.x;
    const int tx = bid % tiles; bid /= tiles;
    const int ty = bid % tiles;
    const int n = bid / tiles;
    const int y = ty * 16 + (tid >> 4), x = tx * 16 + (tid & 15);
    const float* in_n = in + (size_t)n * S * S * C;
    float4 acc = make_float4(bias[0], bias[1], bias[2], bias[3]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        if (iy < 0 || iy >= S || ix < 0 || ix >= S) continue;
        const float4* p = reinterpret_cast<const float4*>(in_n + ((size_t)iy * S + ix) * C);
#pragma unroll 4
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const float4 v = p[c4];
            const float4 w0 = s_w[tap * C + c4 * 4 + 0], w1 = s_w[tap * C + c4 * 4 + 1];
            const float4 w2 = s_w[tap * C + c4 * 4 + 2], w3 = s_w[tap * C + c4 * 4 + 3];
            acc.x = fmaf(v.x, w0.x, acc.x); acc.y = fmaf(v.x, w0.y, acc.y); acc.z = fmaf(v.x, w0.z, acc.z); acc.w = fmaf(v.x, w0.w, acc.w);
            acc.x = fmaf(v.y, w1.x, acc.x); acc.y = fmaf(v.y, w1.y, acc.y); acc.z = fmaf(v.y, w1.z, acc.z); acc.w = fmaf(v.y, w1.w, acc.w);
            acc.x = fmaf(v.z, w2.x, acc.x); acc.y = fmaf(v.z, w2.y, acc.y); acc.z = fmaf(v.z, w2.z, acc.z); acc.w = fmaf(v.z, w2.w, acc.w);
            acc.x = fmaf(v.w, w3.x, acc.x); acc.y = fmaf(v.w, w3.y, acc.y); acc.z = fmaf(v.w, w3.z, acc.z); acc.w = fmaf(v.w, w3.w, acc.w);
        }
    }
    out[((size_t)n * S + y) * S + x] = acc;
}

hipError_t launch_dec_out(hipStream_t st, const float* in, const float* wk, const float* bias, float* out,
                          int N, int S, int C)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
    const int tiles = S / 16;
    if (C == 64)
        hipLaunchKernelGGL((dec_out_kernel<64>), dim3(N * tiles * tiles), dim3(256), 0, st, in,
                           reinterpret_cast<const float4*>(wk), bias, reinterpret_cast<float4*>(out), S, tiles);
    else if (C == 32)
        hipLaunchKernelGGL((dec_out_kernel<32>), dim3(N * tiles * tiles), dim3(256), 0, st, in,
                           reinterpret_cast<const float4*>(wk), bias, reinterpret_cast<float4*>(out), S, tiles);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// =========================================================================================
// Split-precision variant of the stride-1 tile conv: fp32 operands are split on the fly into
// fp16 (hi, lo) pairs, v = hi + lo with |lo| <= ulp_f16(v), and the product is evaluated as
//     a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi        (the dropped a_lo*w_lo term is ~2^-22 |a*w|)
// with three v_mfma_f32_32x32x16_f16 per K=16 step, fp32 accumulation.  16x the fp32-MFMA rate / 3 passes
// = 5.3x, at fp32-class accuracy (measured through the whole training step: ELBO 5e-7, gradients 1e-5 relative,
// see DESIGN.md section 4).  Weights are pre-split and pre-scaled by a power of two (their lo parts would be fp16
// subnormals otherwise); activations are split while being staged into LDS.
//   wpk16[chunk][tap][term hi/lo][kh][co] = 8 fp16 (channels chunk*16 + kh*8 .. +7) * wscale
// =========================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// max |w| of a tensor -> meta[0] = scale (power of two with max*scale in [2^12, 2^13)), meta[1] = 1/scale
__global__ __launch_bounds__(1024)
void weight_scale_kernel(const float* __restrict__ w, int n, float* __restrict__ meta)
{
    __shared__ float s_red[16];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    int i = threadIdx.x;
    for (; i + 3072 < n; i += 4096) {
        m0 = fmaxf(m0, fabsf(w[i])); m1 = fmaxf(m1, fabsf(w[i + 1024]));
        m2 = fmaxf(m2, fabsf(w[i + 2048])); m3 = fmaxf(m3, fabsf(w[i + 3072]));
    }
    for (; i < n; i += 1024) m0 = fmaxf(m0, fabsf(w[i]));
    const float m = wave_max_f32(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = 0.f;
        for (int k = 0; k < 16; ++k) mx = fmaxf(mx, s_red[k]);
        int e = 0;
        if (mx > 0.f && isfinite(mx)) { frexpf(mx, &e); }          // mx = f * 2^e, f in [0.5, 1)
        const float scale = ldexpf(1.f, 13 - e);                    // mx * scale in [2^12, 2^13)
        meta[0] = (mx > 0.f && isfinite(mx)) ? scale : 1.f;
        meta[1] = 1.f / meta[0];
    }
}

__global__ void pack_conv_weights_f16_kernel(const float* __restrict__ src, int O, int I, int cin, int cout, int tflip,
                                             const float* __restrict__ meta, _Float16* __restrict__ dst)
{
    const float scale = meta[0];
    const size_t total = pack_f16_total(cin, cout);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x)
        dst[idx] = pack_f16_element(src, O, I, cout, tflip, scale, idx);
}

hipError_t launch_pack_conv_weights_f16(hipStream_t st, const float* src, int O, int I, int cin, int cout, int tflip,
                                        float* meta, void* dst)
{
    hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(1024), 0, st, src, O * I * 9, meta);
    const size_t total = (size_t)(cin / 16) * 9 * 2 * 2 * cout * 8;
    hipLaunchKernelGGL(pack_conv_weights_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, O, I,
                       cin, cout, tflip, meta, (_Float16*)dst);
    return hipGetLastError();
}


#ifdef IODINE_TILE_PROF
__device__ unsigned g_tile_prof[TP_MAXBLK * 8];
#endif

// TH = tile height: 16 (default: 16x16 tile, 64 px per wave, two blocks per CU) or 8 (conv_variant = 5: 8x16 tile, 32 px per
// wave, <= 168 VGPR and 51 KB LDS -> THREE blocks per CU at twice the weight staging per pixel)
template <int CIN, int COUT, int EPI, int TH = 16>
__global__ __launch_bounds__(256, TH == 16 ? 2 : 3)
void conv3x3_tile_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                               const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                               int S, int tiles, int rev)
{
    constexpr int NCHUNK = CIN / 16;
    constexpr int NT = COUT / 32;
    static_assert(TH == 16 || TH == 8, "tile height");
    constexpr int MT = TH / 8;                           // 32-pixel row-blocks per wave (a wave owns TH / 4 tile rows)
    constexpr int HALO = 18, NPX = HALO * (TH + 2);
    constexpr int PXS = 80;                              // bytes per staged pixel: 32 hi + 32 lo + 16 pad
    constexpr int IN_BYTES = (NPX + 1) * PXS;            // +1 pixel: dump slot for idle lanes
    constexpr int W_U4 = 9 * 2 * 2 * COUT;               // uint4 (8 x fp16) per chunk
    constexpr int NIN = (NPX * 4 + 255) / 256;           // float4 loads per thread per chunk (6)
    constexpr int NW = (W_U4 + 255) / 256;               // uint4 loads per thread per chunk

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* s_in = smem_b;
    uint4* s_w = reinterpret_cast<uint4*>(smem_b + IN_BYTES);
    float* s_max = reinterpret_cast<float*>(smem_b + IN_BYTES + W_U4 * 16);

    TP_DECL;
    using std::integral_constant;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int prow = li >> 4, pcol = li & 15;

    int bid = rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;    // zig-zag launch order, see conv_f16x3()
    const int tx = bid % tiles; bid /= tiles;
    const int tiles_y = tiles * (16 / TH);
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;

    // Global traffic goes through RAW BUFFER loads issued as inline asm:
    //   * one descriptor per tensor (input: this block's slot-image, num_records = one image; packed weights), byte
    //     offsets in a VGPR, the chunk offset in the scalar offset operand -> no 64-bit address arithmetic per load;
    //   * halo pixels outside the image and idle lanes carry the offset 0x80000000 (>= num_records): the hardware returns
    //     0, there is no select after the load;
    //   * hipcc does not count asm loads, so the waits below are explicit counted vmcnt (loads return in order).
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    auto make_rsrc = [&](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)base;
        i32x4_ r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));          // stride 0, no swizzle
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;                                                         // raw dword buffer (gfx9 / CDNA)
        return r;
    };
    const i32x4_ rsrc_in = make_rsrc(in + (size_t)n * S * S * CIN, (unsigned)(S * S * CIN * 4));
    const i32x4_ rsrc_w = make_rsrc(wpk, (unsigned)(NCHUNK * W_U4 * 16));
// (IOD_SGPR_SETTLE before a group of asm memory instructions: an SGPR written by the SALU needs 5 wait states before a
//  VMEM instruction reads it, and hipcc's hazard recognizer does not look inside inline asm; naming the SGPR operands as
//  inputs of the s_nop puts their producers in front of it)
#define IOD_BLOAD4(dst, voff, rsrc, soff) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory")
#define IOD_SGPR_SETTLE(rsrc, soff) asm volatile("s_nop 4" :: "s"(rsrc), "s"(soff) : "memory")
    unsigned goff[NIN];                                  // byte offsets of this thread's halo float4s inside the slot-image
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int idx = tid + k * 256;
        const int px = idx >> 2, cq = idx & 3;
        const int gy = ty * TH - 1 + px / HALO, gx = tx * 16 - 1 + px % HALO;
        const bool ok = idx < NPX * 4 && gy >= 0 && gy < S && gx >= 0 && gx < S;
        goff[k] = ok ? (unsigned)(((gy * S + gx) * CIN + cq * 4) * 4) : 0x80000000u;
#ifdef IODINE_ABL_NOINLOAD
        goff[k] = 0x80000000u;
#endif
    }
    unsigned woff[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) woff[k] = tid + k * 256 < W_U4 ? (unsigned)((tid + k * 256) * 16) : 0x80000000u;
#ifdef IODINE_ABL_NOWLOAD
#pragma unroll
    for (int k = 0; k < NW; ++k) woff[k] = 0x80000000u;
#endif

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // The chunk loop is fully unrolled (straight-line code keeps hipcc's vmcnt waits COUNTED: a loop back-edge makes it
    // drain vmcnt(0), i.e. wait for the prefetch it has just issued).  Input chunks are fetched two steps ahead into two
    // alternating register sets, the (L2-resident) packed weights one step ahead.
    f32x4 rinA[NIN], rinB[NIN];                          // native vectors: they are inline-asm operands
    u32x4_ rw[NW];
    auto prefetch_in = [&](int chunk, f32x4 (&rin)[NIN]) {
        const int soff = chunk * 64;                         // 16 channels
        IOD_SGPR_SETTLE(rsrc_in, soff);
#pragma unroll
        for (int k = 0; k < NIN; ++k) IOD_BLOAD4(rin[k], goff[k], rsrc_in, soff);
    };
    auto prefetch_w = [&](int chunk) {
        const int soff = chunk * (W_U4 * 16);
        IOD_SGPR_SETTLE(rsrc_w, soff);
#pragma unroll
        for (int k = 0; k < NW; ++k) IOD_BLOAD4(rw[k], woff[k], rsrc_w, soff);
    };
    // wait until at most `nc` younger loads are outstanding; the registers are named as read-write operands so that no
    // use of the loaded values can be scheduled above the wait
    auto vm_wait = [&](auto nc, f32x4 (&rin)[NIN]) {
        constexpr int nleft = decltype(nc)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(nleft) : "memory");
#pragma unroll
        for (int k = 0; k < NIN; ++k) asm volatile("" : "+v"(rin[k]));
#pragma unroll
        for (int k = 0; k < NW; ++k) asm volatile("" : "+v"(rw[k]));
        __builtin_amdgcn_sched_barrier(0);
    };
    // Block-local dynamic range: before a chunk is split into fp16 (hi, lo) it is multiplied by a power of two
    // chosen from the chunk tile's max |x| (so that small-magnitude tensors such as gradients keep their lo parts
    // out of the fp16 subnormal range); the accumulators are rescaled (exactly) when the scale changes.
    float cur_scale = 1.f;
    auto commit = [&](const f32x4 (&rin)[NIN]) -> float {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const f32x4 v = rin[k];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        m = wave_max_f32(m);
        TP_STAMP(1);                                         // [1] wait for the chunk's global loads + max
        if (lane == 0) s_max[wv] = m;
        __syncthreads();                                     // also: every wave is done reading the previous chunk
        TP_STAMP(2);                                         // [2] barrier 1
        const float mb = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        // power-of-two scale with hysteresis (tile_scale: keep the current one while max * scale stays in [2^9, 2^14.5),
        // else renormalise to [2^12, 2^13)): most chunks of a tile then share a scale and the 64 accumulator multiplies of
        // a rescale are rare
        const float scale = tile_scale(mb, cur_scale);
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int idx = tid + k * 256;
            const int px = idx < NPX * 4 ? idx >> 2 : NPX, cq = idx & 3;          // idle lanes write the dump slot
            f32x4 v = rin[k];
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            // hi = v truncated to fp16 precision (mask the 13 low mantissa bits: exact in fp16 for the scaled range),
            // lo = v - hi (exact in fp32); both packed with v_cvt_pkrtz (two values per instruction)
            typedef __fp16 h2 __attribute__((ext_vector_type(2)));
            const h2 h01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), h23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);      // (rounding toward zero = the 11 leading bits)
            const h2 l01 = __builtin_amdgcn_cvt_pkrtz(v.x - (float)h01.x, v.y - (float)h01.y), l23 = __builtin_amdgcn_cvt_pkrtz(v.z - (float)h23.x, v.w - (float)h23.y);   // v_fma_mix_f32
            uint2 hi, lo;
            __builtin_memcpy(&hi.x, &h01, 4); __builtin_memcpy(&hi.y, &h23, 4);
            __builtin_memcpy(&lo.x, &l01, 4); __builtin_memcpy(&lo.y, &l23, 4);
            *reinterpret_cast<uint2*>(s_in + px * PXS + cq * 8) = hi;
            *reinterpret_cast<uint2*>(s_in + px * PXS + 32 + cq * 8) = lo;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int idx = tid + k * 256;
            if (idx < W_U4) s_w[idx] = make_uint4(rw[k].x, rw[k].y, rw[k].z, rw[k].w);
        }
        TP_STAMP(3);                                         // [3] scale, split, LDS writes
        __syncthreads();
        TP_STAMP(4);                                         // [4] barrier 2
        return scale;
    };
    auto rescale = [&](float new_scale) {
        if (new_scale != cur_scale) {                        // block-uniform
            const float r = new_scale / cur_scale;           // exact: both are powers of two
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[mt][nt][q] *= r;
            cur_scale = new_scale;
        }
    };

    // Fragment reads are inline-asm ds_read_b128 (invisible to hipcc's wait-count pass, which otherwise drains
    // lgkmcnt(0) - including the reads just issued for the NEXT tap - in front of every MFMA group).  LDS returns in
    // order, so a counted wait after issuing the next tap's 4 + 2*NT reads retires exactly the current tap's fragments.
    struct Frag { f16x8 ah[MT], al[MT], bh[NT], bl[NT]; };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_b;
    const unsigned a_addr0 = lds_base + (((TH / 4) * wv + prow) * HALO + pcol) * PXS + kh * 16;
    const unsigned a_addr1 = a_addr0 + 2 * HALO * PXS;
    const unsigned b_addr = lds_base + IN_BYTES + (kh * COUT + li) * 16;
#define IOD_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto LOADF = [&](auto tapc, Frag& f) {
        constexpr int tap = decltype(tapc)::value;
        constexpr int aoff = ((tap / 3) * HALO + (tap % 3)) * PXS;
        constexpr int boff = tap * 4 * COUT * 16;
#ifdef IODINE_ABL_NOLDSREAD
        asm volatile("" : "+v"(f.ah[0]), "+v"(f.al[0]), "+v"(f.ah[MT - 1]), "+v"(f.al[MT - 1]), "+v"(f.bh[0]), "+v"(f.bl[0]));
        if constexpr (NT == 2) asm volatile("" : "+v"(f.bh[1]), "+v"(f.bl[1]));
        return;
#endif
        IOD_DSR128(f.ah[0], a_addr0, aoff);
        IOD_DSR128(f.al[0], a_addr0, aoff + 32);
        if constexpr (MT == 2) {
            IOD_DSR128(f.ah[MT - 1], a_addr1, aoff);
            IOD_DSR128(f.al[MT - 1], a_addr1, aoff + 32);
        }
        IOD_DSR128(f.bh[0], b_addr, boff);
        IOD_DSR128(f.bl[0], b_addr, boff + 2 * COUT * 16);
        if constexpr (NT == 2) {
            IOD_DSR128(f.bh[1], b_addr, boff + 512);
            IOD_DSR128(f.bl[1], b_addr, boff + 2 * COUT * 16 + 512);
        }
    };
#undef IOD_DSR128
    auto MMA = [&](const Frag& f) {
#ifdef IODINE_ABL_NOMFMA
        asm volatile("" :: "v"(f.ah[0]), "v"(f.al[0]), "v"(f.ah[MT - 1]), "v"(f.al[MT - 1]), "v"(f.bh[0]), "v"(f.bl[0]));
        if constexpr (NT == 2) asm volatile("" :: "v"(f.bh[1]), "v"(f.bl[1]));
        return;
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[nt], f.al[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[nt], f.ah[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[nt], f.ah[mt], acc[mt][nt], 0, 0, 0);
    };
#define IOD_STEP(T, FCUR, FNEXT)                                                                  \
    if constexpr (T + 1 < 9) LOADF(integral_constant<int, (T + 1 < 9 ? T + 1 : 8)>{}, FNEXT);       \
    if constexpr (T + 1 < 9) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(2 * MT + 2 * NT) : "memory");   \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MMA(FCUR);                                                                                      \
    __builtin_amdgcn_sched_barrier(0);

    auto compute = [&]() {
        TP_STAMP(5);                                         // [5] issue of the next prefetches (between commit and compute)
        Frag f0, f1;
#ifdef IODINE_ABL_NOLDSREAD
        f0 = Frag{}; f1 = Frag{};
#endif
        LOADF(integral_constant<int, 0>{}, f0);
        IOD_STEP(0, f0, f1) IOD_STEP(1, f1, f0) IOD_STEP(2, f0, f1) IOD_STEP(3, f1, f0) IOD_STEP(4, f0, f1)
        IOD_STEP(5, f1, f0) IOD_STEP(6, f0, f1) IOD_STEP(7, f1, f0) IOD_STEP(8, f0, f1)
        TP_STAMP(6);                                         // [6] 9 taps of LDS fragment reads + MFMA
    };
    // data-gradient form: the ELU' operand (the saved activation of the layer below, one float4 per accumulator quad) is
    // requested right after the LAST chunk has been staged - its input / weight registers are free by then - and arrives
    // under the last 108 MFMAs instead of being waited for in the epilogue
    // Output-side lane mapping (see the epilogue): lane = (pixel pl of PPI, 16-byte channel segment seg); instruction j
    // covers pixels j*PPI .. of the wave's 64.  The ELU' operand is fetched in exactly this shape - whole pixels, 1 KB
    // per instruction - and applied AFTER the accumulators have been transposed through LDS.
    constexpr int SEGS = COUT / 4;                           // float4 segments per pixel (16 / 8)
    constexpr int PPI = 64 / SEGS;                           // pixels per load / store instruction (4 / 8)
    constexpr int NEP = 32 * MT / PPI;                       // instructions per wave tile (16 / 8; half of it at TH = 8)
    const int seg = lane % SEGS, pl = lane / SEGS;
    const unsigned vbase = (unsigned)((((ty * TH + (TH / 4) * wv) * S + tx * 16 + pl) * COUT + seg * 4) * 4);
    f32x4 ax[NEP];
    auto prefetch_aux = [&]() {
        if constexpr (EPI == EPI_MUL_ELUGRAD || EPI == EPI_L0ROWS) {
            const i32x4_ rsrc_aux = make_rsrc(aux + (size_t)n * S * S * COUT, (unsigned)(S * S * COUT * 4));
#pragma unroll
            for (int j = 0; j < NEP; ++j) {
                const int soff = (((j * PPI) / 16) * S + (j * PPI) % 16) * COUT * 4;
                IOD_SGPR_SETTLE(rsrc_aux, soff);
                IOD_BLOAD4(ax[j], vbase, rsrc_aux, soff);
            }
        }
    };
    TP_STAMP(0);                                             // [0] block start: index arithmetic
    prefetch_in(0, rinA);
    prefetch_w(0);
    if (NCHUNK > 1) prefetch_in(1, rinB);
    vm_wait(integral_constant<int, (NCHUNK > 1 ? NIN : 0)>{}, rinA);      // chunk 0 + weights 0 arrived (chunk 1 may be in flight)
    cur_scale = commit(rinA);                                // chunk 0 staged
    if (NCHUNK > 1) prefetch_w(1);
    if (NCHUNK > 2) prefetch_in(2, rinA);
    compute();                                               // chunk 0
    if constexpr (NCHUNK > 1) {
        vm_wait(integral_constant<int, (NCHUNK > 2 ? NIN : 0)>{}, rinB);
        rescale(commit(rinB));                               // chunk 1 staged
        if (NCHUNK > 2) prefetch_w(2);
        if (NCHUNK > 3) prefetch_in(3, rinB);
        if (NCHUNK == 2) prefetch_aux();
        compute();                                           // chunk 1
    }
    if constexpr (NCHUNK > 2) {
        vm_wait(integral_constant<int, (NCHUNK > 3 ? NIN : 0)>{}, rinA);
        rescale(commit(rinA));
        if (NCHUNK > 3) prefetch_w(3);
        if (NCHUNK == 3) prefetch_aux();
        compute();                                           // chunk 2
    }
    if constexpr (NCHUNK > 3) {
        vm_wait(integral_constant<int, 0>{}, rinB);
        rescale(commit(rinB));
        prefetch_aux();
        compute();                                           // chunk 3
    }
    static_assert(NCHUNK <= 4, "chunk schedule is unrolled for at most 64 input channels");
#undef IOD_STEP

    const float inv_ws = wmeta[1] / cur_scale;
    // The MFMAs are issued as (weights, activations): D = W^T A^T, so a lane's accumulator rows are CHANNELS - lane
    // (li, kh) holds pixel li of the 32-pixel tile and channels 8g + 4kh .. +3 in registers 4g .. 4g+3: the epilogue
    // moves float4s of 4 consecutive channels.  Stores (and the ELU' operand of the data-gradient form) are raw buffer
    // operations on the slot-image: one byte offset per 32-pixel tile and lane, channel group in the scalar offset.
    static_assert(EPI == EPI_BIAS_ELU || EPI == EPI_MUL_ELUGRAD || EPI == EPI_L0ROWS, "the C -> 4 output conv has its own GEMM-form kernel");
    {
        const i32x4_ rsrc_out = make_rsrc(out + (size_t)n * S * S * COUT, (unsigned)(S * S * COUT * 4));
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_BIAS_ELU) {
            const float4 t = *reinterpret_cast<const float4*>(bias + seg * 4);
            b4 = f32x4{t.x, t.y, t.z, t.w};
            asm volatile("" : "+v"(b4));                     // (pinned before the asm stores, see the direct form)
        }
        if constexpr (EPI == EPI_MUL_ELUGRAD || EPI == EPI_L0ROWS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < NEP; ++j) asm volatile("" : "+v"(ax[j]));
            __builtin_amdgcn_sched_barrier(0);
        }
        // The finished tile goes through LDS once more so that the global stores are CONTIGUOUS: a lane's accumulator
        // float4 is 16 bytes of one pixel (256-byte pixel stride: a direct store instruction touches 32 cache lines with
        // 32 bytes each), whereas after the transposition every store instruction writes four whole pixels = 1 KB.
        // Each wave owns a [64 pixels][COUT + 4 floats] region (the pad keeps the column-shaped writes off one bank).
        constexpr int EPS = (COUT + 4) * 4;                      // bytes per staged pixel
        __syncthreads();                                         // every wave is done with the staging buffers
        unsigned char* s_ep = smem_b + wv * (32 * MT) * EPS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<f32x4*>(s_ep + (mt * 32 + li) * EPS + (nt * 32 + 8 * g4 + 4 * kh) * 4) =
                        f32x4{acc[mt][nt][4 * g4] * inv_ws, acc[mt][nt][4 * g4 + 1] * inv_ws,
                              acc[mt][nt][4 * g4 + 2] * inv_ws, acc[mt][nt][4 * g4 + 3] * inv_ws};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // instruction j: pixels j*PPI .. of the wave's 64 (tile row (j*PPI)/16, columns (j*PPI)%16 ..), bias + ELU or the
        // ELU' factor applied in this layout (a lane keeps ONE float4 of bias: its channel segment never changes)
        // EPI_L0ROWS: per tile row the left-border / interior / right-border column sums instead of the pixels
        f32x4 rsum[EPI == EPI_L0ROWS ? TH / 4 : 1][3];
        if constexpr (EPI == EPI_L0ROWS) {
#pragma unroll
            for (int rr = 0; rr < TH / 4; ++rr)
#pragma unroll
                for (int c = 0; c < 3; ++c) rsum[rr][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NEP; ++j) {
            const int pq = j * PPI;
            f32x4 v = *reinterpret_cast<const f32x4*>(s_ep + (pq + pl) * EPS + seg * 16);
            if constexpr (EPI == EPI_BIAS_ELU) {
                v = f32x4{elu1_fast(v.x + b4.x), elu1_fast(v.y + b4.y), elu1_fast(v.z + b4.z), elu1_fast(v.w + b4.w)};
            } else {
                const f32x4 a4 = ax[j];
                v.x *= elu1_grad_from_out(a4.x); v.y *= elu1_grad_from_out(a4.y);
                v.z *= elu1_grad_from_out(a4.z); v.w *= elu1_grad_from_out(a4.w);
            }
            if constexpr (EPI == EPI_L0ROWS) {
                const int gx = tx * 16 + pq % 16 + pl;
                const float wl = gx == 0 ? 1.f : 0.f, wr = gx == S - 1 ? 1.f : 0.f, wm = 1.f - wl - wr;
                rsum[pq / 16][0] += v * wl; rsum[pq / 16][1] += v * wm; rsum[pq / 16][2] += v * wr;
            } else {
                const int soff = ((pq / 16) * S + pq % 16) * COUT * 4;
                // (the s_nops cover two hazards hipcc's recognizer cannot see inside inline asm: an SGPR written by the
                //  SALU needs 5 wait states before a VMEM instruction reads it, and a VALU write of the store's data VGPRs
                //  has to wait for the store to have read them)
#ifdef IODINE_ABL_NOSTORE
                asm volatile("" :: "v"(v), "v"(vbase), "s"(rsrc_out), "s"(soff) : "memory");
#else
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v), "v"(vbase), "s"(rsrc_out), "s"(soff) : "memory");
#endif
            }
        }
        if constexpr (EPI == EPI_L0ROWS) {
            // lanes (pl, seg) -> sum over the PPI pixel lanes through the (now free) tile region, one tile row at a time:
            // out = rows_p[n][y][tx][3][COUT]
            f32x4* s_rs = reinterpret_cast<f32x4*>(s_ep);
            float* rows_p = out;
#pragma unroll
            for (int rr = 0; rr < TH / 4; ++rr) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int c = 0; c < 3; ++c) s_rs[lane * 3 + c] = rsum[rr][c];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 3 * SEGS) {
                    const int c = lane / SEGS, sg = lane % SEGS;
                    f32x4 t = s_rs[sg * 3 + c];
#pragma unroll
                    for (int q = 1; q < PPI; ++q) t += s_rs[(q * SEGS + sg) * 3 + c];
                    const int gy = ty * TH + (TH / 4) * wv + rr;
                    *reinterpret_cast<f32x4*>(rows_p + ((((size_t)n * S + gy) * tiles + tx) * 3 + c) * COUT + sg * 4) = t;
                }
            }
        }
    }
    TP_STAMP(7);                                             // [7] epilogue (issue of the stores)
    TP_FLUSH(g_tile_prof);
#undef IOD_BLOAD4
#undef IOD_SGPR_SETTLE
}

template <int CIN, int COUT, int EPI, int TH>
static hipError_t launch_tile_f16x3_inst(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                         const float* bias, const float* aux, float* out, int N, int S, int rev)
{
    constexpr size_t lds_stage = (size_t)(18 * (TH + 2) + 1) * 80 + (size_t)9 * 2 * 2 * COUT * 16 + 16;
    constexpr size_t lds_epi = (size_t)4 * (TH * 4) * (COUT + 4) * 4;              // output tile, transposed for contiguous stores
    constexpr size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    static std::atomic<unsigned> attr_devs{0};                             // devices this instance is configured on
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_tile_f16x3_kernel<CIN, COUT, EPI, TH>, (int)lds, attr_devs); e != hipSuccess) return e;
    const int tiles = S / 16;
    hipLaunchKernelGGL((conv3x3_tile_f16x3_kernel<CIN, COUT, EPI, TH>), dim3(N * tiles * tiles * (16 / TH)), dim3(256), lds, st, in,
                       reinterpret_cast<const uint4*>(wpk), wmeta, bias, aux, out, S, tiles, rev);
#ifdef IODINE_TILE_PROF
    {
        const int nb = std::min(N * tiles * tiles * (16 / TH), TP_MAXBLK);
        std::vector<unsigned> hp((size_t)nb * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_tile_prof), hp.size() * sizeof(unsigned));
        static const char* names[8] = {"start", "load-wait+max", "barrier1", "split+lds-write", "barrier2", "prefetch-issue",
                                       "taps(lds-read+mfma)", "epilogue"};
        double sum[8] = {0}, tot = 0;
        for (int b2 = 0; b2 < nb; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[(size_t)b2 * 8 + i];
        for (int i = 0; i < 8; ++i) tot += sum[i] / nb;
        fprintf(stderr, "[tile prof] memtime ticks per block (wave 0), total %.0f:", tot);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f |", names[i], sum[i] / nb);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

hipError_t launch_conv3x3_tile_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                     const float* bias, const float* aux, float* out, int N, int S, int cin, int cout,
                                     int epi, int rev)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
#define T16_CASE(CI, CO, EP) \
    if (cin == CI && cout == CO && epi == EP) return launch_tile_f16x3_inst<CI, CO, EP, 16>(st, in, wpk, wmeta, bias, aux, out, N, S, rev);
    T16_CASE(64, 64, EPI_BIAS_ELU) T16_CASE(64, 64, EPI_MUL_ELUGRAD) T16_CASE(64, 64, EPI_L0ROWS)
    T16_CASE(32, 32, EPI_BIAS_ELU) T16_CASE(32, 32, EPI_MUL_ELUGRAD) T16_CASE(32, 32, EPI_L0ROWS)
#undef T16_CASE
    return hipErrorInvalidValue;
}

// =========================================================================================
// Decoder output conv C -> 4 in GEMM form (lib/modeling/iodine.py:422,435).  With only 4 output channels the 3x3 conv
// is cheaper as   P[q][tap*4 + co] = sum_ci a[q][ci] * W[co][ci][tap]      (one [pixels x C] . [C x 36] GEMM, no taps in K)
// followed by     out[p][co] = bias[co] + sum_tap P[p + tap - 1][tap*4 + co]   (9 LDS float4 reads per pixel):
// 4.5x fewer MFMAs than padding N = 4 up to a 32-wide tile in the direct form.  Split-fp16 (3 MFMA) arithmetic and the
// block-local power-of-two scaling are those of conv3x3_tile_f16x3_kernel.
//   block = 256 threads, 16x16 output pixels, 18x18 halo (11 row-blocks of 32 pixels over 4 waves), K in chunks of 16.
//   wpk: [chunk][term hi/lo][kh][n 0..63][8 fp16], n = tap*4 + co (n >= 36 zero)
// =========================================================================================
__global__ void pack_dec_out_gemm_kernel(const float* __restrict__ w /*[4][C][3][3]*/, int C, const float* __restrict__ meta,
                                         _Float16* __restrict__ dst)
{
    const float scale = meta[0];
    const int total = (int)pack_out_gemm_total(C);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
        dst[idx] = pack_out_gemm_element(w, C, scale, (size_t)idx);          // (pack_bodies.h: shared with the batched form)
}

hipError_t launch_pack_dec_out_gemm(hipStream_t st, const float* w, int C, float* meta, void* dst)
{
    hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(1024), 0, st, w, 4 * C * 9, meta);
    hipLaunchKernelGGL(pack_dec_out_gemm_kernel, dim3(16), dim3(256), 0, st, w, C, meta, (_Float16*)dst);
    return hipGetLastError();
}
