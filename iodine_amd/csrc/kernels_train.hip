// Training-only kernels: weight gradients of the conv stacks (fp32 MFMA, reduction over pixels),
// parameter gradients of the spatial-broadcast layer, and the small dense algebra of the
// refinement head's backward (LSTM BPTT).
//
// They replace what autograd does for the outer ``loss.backward()`` (lib/engine/train.py:63) on the
// graph built by IODINE.forward (lib/modeling/iodine.py:115-158).  Because every ELBO_i reaches the
// decoder parameters only through decoder pass i, and the inner backward of iteration i already
// produced d(B*ELBO_i)/d(pre-activations) for every decoder layer, the decoder weight gradients are
// accumulated during the forward loop with the factor -w_i/B (w_i = (i+1)/(T+1), iodine.py:151-153);
// the refinement network is back-propagated through time afterwards (inputs are detached,
// iodine.py:343; lambda is detached before the additive update, iodine.py:642-643).
#include "common.h"
#include <type_traits>
#include <utility>
#include <vector>
#include <cstdio>

// =========================================================================================
// stride-1 3x3 weight gradient, LDS-tiled:  dW[tap][ci][co] = sum_{n,p} a[n, p+tap, ci] * d[n, p, co]
//   GEMM view per tap: M = ci, N = co, K = pixels.  Block = 4 waves, persistent over 8x16-pixel tiles;
//   wave -> (ci half, co half, pixel split); each wave keeps the 9 taps' 32x32 accumulators (144 regs).
//   Partial sums go to part[(block*KS + ks)][tap][ci][co_pad]; a second kernel reduces them in fixed order.
// =========================================================================================
template <int CI, int NCO>
__global__ __launch_bounds__(256, 2)
void conv3x3_wgrad_tile_kernel(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ part,
                               float* __restrict__ part_b, int S, int ntiles, int tiles_x, int tiles_y)
{
    constexpr int MT = CI / 32;
    constexpr int NTT = (NCO + 31) / 32;
    constexpr int KS = 4 / (MT * NTT);
    constexpr int NCOP = NTT * 32;
    constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2;
    constexpr int A4 = CI / 4, D4 = NCO / 4;
    constexpr int PXW = (TH * TW) / KS;                 // pixels per wave per tile

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a = smem;                                   // HH*HW*CI
    float* s_d = smem + HH * HW * CI;                    // TH*TW*NCO

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, li = lane & 31;
    const int mi = wv % MT, ni = (wv / MT) % NTT, ks = wv / (MT * NTT);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        const float* a_n = a + (size_t)n * S * S * CI;
        const float* d_n = d + (size_t)n * S * S * NCO;
        __syncthreads();
        for (int idx = tid; idx < HH * HW * A4; idx += 256) {
            const int px = idx / A4, c4 = idx % A4;
            const int gy = y0 + px / HW, gx = x0 + px % HW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < S && gx >= 0 && gx < S)
                v = *reinterpret_cast<const float4*>(a_n + ((size_t)gy * S + gx) * CI + c4 * 4);
            *reinterpret_cast<float4*>(s_a + px * CI + c4 * 4) = v;
        }
        for (int idx = tid; idx < TH * TW * D4; idx += 256) {
            const int px = idx / D4, c4 = idx % D4;
            const int gy = ty * TH + px / TW, gx = tx * TW + px % TW;
            const float4 v = *reinterpret_cast<const float4*>(d_n + ((size_t)gy * S + gx) * NCO + c4 * 4);
            *reinterpret_cast<float4*>(s_d + px * NCO + c4 * 4) = v;
            bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
        }
        __syncthreads();
#pragma unroll 2
        for (int s = 0; s < PXW / 2; ++s) {
            const int px = ks * PXW + 2 * s + half;
            const int r = px / TW, c = px % TW;
            float bval = 0.f;
            if (NCO >= 32 || li < NCO) bval = s_d[px * NCO + ni * 32 + li];
            const float* ap = s_a + (r * HW + c) * CI + mi * 32 + li;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float aval = ap[((tap / 3) * HW + (tap % 3)) * CI];
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bval, acc[tap], 0, 0, 0);
            }
        }
    }

    // partial dW: rows = ci (accumulator rows), cols = co (lane & 31)
    float* pw = part + ((size_t)(blockIdx.x * KS + ks) * 9) * CI * NCOP;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            pw[((size_t)tap * CI + ci) * NCOP + ni * 32 + li] = acc[tap][r];
        }
    // partial bias gradient: thread's channel group is fixed (256 % D4 == 0)
    __syncthreads();
    float4* s_red = reinterpret_cast<float4*>(smem);
    s_red[tid] = bsum;
    __syncthreads();
    if (tid < D4) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = tid; j < 256; j += D4) { const float4 v = s_red[j]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
        *reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * NCO + tid * 4) = t4;
    }
}

int wgrad_tile_blocks(int N, int S) { const int nt = N * (S / 8) * (S / 16); return nt < 512 ? nt : 512; }

template <int CI, int NCO>
static hipError_t launch_wgrad_tile_inst(hipStream_t st, const float* a, const float* d, float* part, float* part_b,
                                         int N, int S, int* nparts, int* ncop)
{
    constexpr int MT = CI / 32, NTT = (NCO + 31) / 32, KS = 4 / (MT * NTT);
    constexpr size_t lds = (size_t)(10 * 18 * CI + 8 * 16 * NCO) * 4;
    static std::atomic<unsigned> attr_devs{0};                             // devices this instance is configured on
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_wgrad_tile_kernel<CI, NCO>, (int)lds, attr_devs); e != hipSuccess) return e;
    const int tiles_x = S / 16, tiles_y = S / 8, ntiles = N * tiles_x * tiles_y;
    const int blocks = wgrad_tile_blocks(N, S);
    hipLaunchKernelGGL((conv3x3_wgrad_tile_kernel<CI, NCO>), dim3(blocks), dim3(256), lds, st, a, d, part, part_b, S,
                       ntiles, tiles_x, tiles_y);
    *nparts = blocks * KS;
    *ncop = NTT * 32;
    return hipGetLastError();
}

hipError_t launch_conv3x3_wgrad_tile(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N,
                                     int S, int ci, int nco, int* nparts, int* ncop, int* nbias_parts)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
    *nbias_parts = wgrad_tile_blocks(N, S);
    if (ci == 64 && nco == 64) return launch_wgrad_tile_inst<64, 64>(st, a, d, part, part_b, N, S, nparts, ncop);
    if (ci == 32 && nco == 32) return launch_wgrad_tile_inst<32, 32>(st, a, d, part, part_b, N, S, nparts, ncop);
    if (ci == 64 && nco == 4) return launch_wgrad_tile_inst<64, 4>(st, a, d, part, part_b, N, S, nparts, ncop);
    if (ci == 32 && nco == 4) return launch_wgrad_tile_inst<32, 4>(st, a, d, part, part_b, N, S, nparts, ncop);
    return hipErrorInvalidValue;
}

// =========================================================================================
// strided 3x3 weight gradient, operands gathered from global memory (refinement stack):
//   dW[tap][ci][co] = sum_m in[n(m), STRIDE*o(m) + tap - 1, ci] * d[m, co],  m over all output pixels.
// =========================================================================================
template <int CIP, int CO, int STRIDE>
__global__ __launch_bounds__(256, 2)
void conv3x3_wgrad_gather_kernel(const float* __restrict__ in, const float* __restrict__ d, float* __restrict__ part,
                                 int M, int IH, int IW, int OH, int OW, int nchunks)
{
    constexpr int MT = (CIP + 31) / 32;
    constexpr int NTT = CO / 32;
    constexpr int KS = 4 / (MT * NTT);
    constexpr int PXW = 128 / KS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, li = lane & 31;
    const int mi = wv % MT, ni = (wv / MT) % NTT, ks = wv / (MT * NTT);
    const int ci = mi * 32 + li;
    const bool ci_ok = ci < CIP;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    constexpr int UNR = 4;                                   // k-steps whose 10 gathers each are issued before any MFMA
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        for (int s0 = 0; s0 < PXW / 2; s0 += UNR) {
            float bval[UNR], aval[UNR][9];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int m = chunk * 128 + ks * PXW + 2 * (s0 + u) + half;
                const bool valid = m < M;
                int t = valid ? m : 0;
                const int ox = t % OW; t /= OW;
                const int oy = t % OH;
                const int n = t / OH;
                bval[u] = valid ? d[(size_t)m * CO + ni * 32 + li] : 0.f;
                const float* in_n = in + (size_t)n * IH * IW * CIP + ci;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int iy = oy * STRIDE + tap / 3 - 1, ix = ox * STRIDE + tap % 3 - 1;
                    float v = 0.f;
                    if (valid && ci_ok && iy >= 0 && iy < IH && ix >= 0 && ix < IW) v = in_n[((size_t)iy * IW + ix) * CIP];
                    aval[u][tap] = v;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval[u][tap], bval[u], acc[tap], 0, 0, 0);
        }
    }
    constexpr int CIPAD = MT * 32;
    float* pw = part + ((size_t)(blockIdx.x * KS + ks) * 9) * CIPAD * CO;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            pw[((size_t)tap * CIPAD + cr) * CO + ni * 32 + li] = acc[tap][r];
        }
}

template <int CIP, int CO, int STRIDE>
static hipError_t launch_wgrad_gather_inst(hipStream_t st, const float* in, const float* d, float* part, int N, int IH,
                                           int IW, int* nparts, int* cipad)
{
    constexpr int MT = (CIP + 31) / 32, NTT = CO / 32, KS = 4 / (MT * NTT);
    const int OH = (IH - 1) / STRIDE + 1, OW = (IW - 1) / STRIDE + 1;
    const int M = N * OH * OW, nchunks = (M + 127) / 128;
    const int blocks = nchunks < 512 ? nchunks : 512;
    hipLaunchKernelGGL((conv3x3_wgrad_gather_kernel<CIP, CO, STRIDE>), dim3(blocks), dim3(256), 0, st, in, d, part, M, IH,
                       IW, OH, OW, nchunks);
    *nparts = blocks * KS;
    *cipad = MT * 32;
    return hipGetLastError();
}

hipError_t launch_conv3x3_wgrad_gather(hipStream_t st, const float* in, const float* d, float* part, int N, int IH,
                                       int IW, int cip, int co, int stride, int* nparts, int* cipad)
{
    if (stride != 2) return hipErrorInvalidValue;
    if (cip == 20 && co == 64) return launch_wgrad_gather_inst<20, 64, 2>(st, in, d, part, N, IH, IW, nparts, cipad);
    if (cip == 64 && co == 64) return launch_wgrad_gather_inst<64, 64, 2>(st, in, d, part, N, IH, IW, nparts, cipad);
    if (cip == 20 && co == 32) return launch_wgrad_gather_inst<20, 32, 2>(st, in, d, part, N, IH, IW, nparts, cipad);
    if (cip == 32 && co == 32) return launch_wgrad_gather_inst<32, 32, 2>(st, in, d, part, N, IH, IW, nparts, cipad);
    return hipErrorInvalidValue;
}

// fixed-order reduction of the partial tiles into an OIHW gradient:
//   dst[co][ci_off + ci][tap] += alpha * sum_b part[b][tap][ci][co]      (ci < I_real, co < O_real)
__global__ __launch_bounds__(256)
void wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int ci_pad, int co_pad, int O_real,
                         int I_real, int I_dst, float alpha, float* __restrict__ dst,
                         const float* __restrict__ part_b, int nb, float* __restrict__ dst_b)
{
    const int total = 9 * ci_pad * co_pad, nblk_w = (total + 255) / 256;
    if ((int)blockIdx.x >= nblk_w) {
        // the blocks behind the weight elements sum the bias partial rows of the same launch (one column each): saves the
        // separate column-sum launch per weight-gradient launch
        __shared__ float s_red[256];
        const int c = blockIdx.x - nblk_w, tid = threadIdx.x;
        float s = 0.f;
        for (int r = tid; r < nb; r += 256) s += part_b[(size_t)r * O_real + c];
        s_red[tid] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        if (tid == 0) dst_b[c] += alpha * s_red[0];
        return;
    }
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int co = e % co_pad, ci = (e / co_pad) % ci_pad, tap = e / (co_pad * ci_pad);
    if (co >= O_real || ci >= I_real) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < nparts; b += 4) {
        s0 += part[(size_t)b * total + e];
        s1 += part[(size_t)(b + 1) * total + e];
        s2 += part[(size_t)(b + 2) * total + e];
        s3 += part[(size_t)(b + 3) * total + e];
    }
    for (; b < nparts; ++b) s0 += part[(size_t)b * total + e];
    float* o = dst + ((size_t)co * I_dst + ci) * 9 + tap;
    *o += alpha * ((s0 + s1) + (s2 + s3));
}

// first stage for many partial tiles: fold[f][e] = sum of the parts f*per .. f*per+per-1 (float4 columns, all of a
// thread's loads independent), so that the layout-changing second stage reads WGRAD_FOLD tiles instead of hundreds
__global__ __launch_bounds__(256)
void wgrad_fold_kernel(const float4* __restrict__ part, int nparts, int per, int total4, float4* __restrict__ fold)
{
    const int e = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (e >= total4) return;
    const int b0 = f * per, b1 = min(nparts, b0 + per);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int b = b0;
    for (; b + 7 < b1; b += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(b + j) * total4 + e];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            s0.x += v[j].x; s0.y += v[j].y; s0.z += v[j].z; s0.w += v[j].w;
            s1.x += v[j + 1].x; s1.y += v[j + 1].y; s1.z += v[j + 1].z; s1.w += v[j + 1].w;
        }
    }
    for (; b < b1; ++b) { const float4 v = part[(size_t)b * total4 + e]; s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w; }
    fold[(size_t)f * total4 + e] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
}

// Round 4: both stages in ONE launch (the two-launch form cost 28 x (7.7 + 5.4 us + a kernel boundary) per cfg3 training step).
// A block owns 64 consecutive elements of the [tap][ci][co] tile: thread (g = tid / 16, q = tid % 16) sums float4 column q of the
// parts g, g + 16, g + 32, ... (eight independent 16-byte loads in flight), the 16 group sums meet in LDS and are added in fixed order
// by the first 64 threads, which also do the layout change into OIHW.  Blocks behind the weight blocks sum the bias partial rows.
__global__ __launch_bounds__(256)
void wgrad_reduce_fused_kernel(const float4* __restrict__ part, int nparts, int ci_pad, int co_pad, int O_real, int I_real, int I_dst,
                               float alpha, float* __restrict__ dst, const float* __restrict__ part_b, int nb, float* __restrict__ dst_b)
{
    const int total = 9 * ci_pad * co_pad, total4 = total / 4, nblk_w = total / 64;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nblk_w) {
        __shared__ float s_red[256];
        const int c = blockIdx.x - nblk_w;
        float s = 0.f;
        for (int r = tid; r < nb; r += 256) s += part_b[(size_t)r * O_real + c];
        s_red[tid] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        if (tid == 0) dst_b[c] += alpha * s_red[0];
        return;
    }
    __shared__ float4 s_g[16][16];
    const int g = tid >> 4, q = tid & 15;
    const float4* col = part + (size_t)blockIdx.x * 16 + q;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int b = g;
    for (; b + 7 * 16 < nparts; b += 8 * 16) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = col[(size_t)(b + 16 * j) * total4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            s0.x += v[j].x; s0.y += v[j].y; s0.z += v[j].z; s0.w += v[j].w;
            s1.x += v[j + 1].x; s1.y += v[j + 1].y; s1.z += v[j + 1].z; s1.w += v[j + 1].w;
        }
    }
    for (; b < nparts; b += 16) { const float4 v = col[(size_t)b * total4]; s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w; }
    s_g[g][q] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (tid < 64) {
        const float* sf = reinterpret_cast<const float*>(&s_g[0][0]);
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j += 2) { t0 += sf[j * 64 + tid]; t1 += sf[(j + 1) * 64 + tid]; }
        const int e = blockIdx.x * 64 + tid;
        const int co = e % co_pad, ci = (e / co_pad) % ci_pad, tap = e / (co_pad * ci_pad);
        if (co < O_real && ci < I_real) {
            float* o = dst + ((size_t)co * I_dst + ci) * 9 + tap;
            *o += alpha * (t0 + t1);
        }
    }
}

// part_b / dst_b (optional): nb bias partial rows [nb][O_real] of the same launch, summed into dst_b with the same alpha
hipError_t launch_wgrad_reduce(hipStream_t st, const float* part, int nparts, int ci_pad, int co_pad, int O_real,
                               int I_real, int I_dst, float alpha, float* dst, float* fold, const float* part_b, int nb,
                               float* dst_b)
{
    IOD_XSKIP(1);
    if (nparts <= 0) return hipSuccess;
    const int total = 9 * ci_pad * co_pad;
    const bool with_b = part_b && dst_b && nb > 0;
    if (nparts >= 16 && total % 64 == 0) {
        hipLaunchKernelGGL(wgrad_reduce_fused_kernel, dim3(total / 64 + (with_b ? O_real : 0)), dim3(256), 0, st, (const float4*)part, nparts,
                           ci_pad, co_pad, O_real, I_real, I_dst, alpha, dst, part_b, nb, dst_b);
        return hipGetLastError();
    }
    if (fold && nparts >= 4 * WGRAD_FOLD && total % 4 == 0) {
        const int per = (nparts + WGRAD_FOLD - 1) / WGRAD_FOLD, nf = (nparts + per - 1) / per;
        hipLaunchKernelGGL(wgrad_fold_kernel, dim3((total / 4 + 255) / 256, nf), dim3(256), 0, st, (const float4*)part,
                           nparts, per, total / 4, (float4*)fold);
        part = fold;
        nparts = nf;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256 + (with_b ? O_real : 0)), dim3(256), 0, st, part, nparts,
                       ci_pad, co_pad, O_real, I_real, I_dst, alpha, dst, part_b, nb, dst_b);
    return hipGetLastError();
}

// dst[c] += alpha * sum_r src[r][c]   (bias gradients, init_mean / init_logvar gradients, partial-bias reduction)
__global__ void colsum_kernel(const float* __restrict__ src, int rows, int cols, int ld, float alpha, float* __restrict__ dst)
{
    __shared__ float s_red[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int r = tid; r < rows; r += 256) s += src[(size_t)r * ld + c];
    s_red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    if (tid == 0) dst[c] += alpha * s_red[0];
}

hipError_t launch_colsum(hipStream_t st, const float* src, int rows, int cols, int ld, float alpha, float* dst)
{
    IOD_XSKIP(1);
    hipLaunchKernelGGL(colsum_kernel, dim3(cols), dim3(256), 0, st, src, rows, cols, ld, alpha, dst);
    return hipGetLastError();
}

// column sums of a tall matrix (rows ~ 1e6, cols <= 256, cols % 4 == 0, contiguous): coalesced float4 streaming,
// one partial row per block, then the small column-sum kernel over the partials (fixed order -> deterministic).
__global__ __launch_bounds__(256)
void colsum_tall_partial_kernel(const float4* __restrict__ src, int rows, int c4, int rows_per_block,
                                float4* __restrict__ partial)
{
    __shared__ float4 s_red[256];
    const int tid = threadIdx.x;
    const int rl = 256 / c4;                               // row lanes
    const int c = tid % c4, r0 = tid / c4;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(rows, rbeg + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < rl)
        for (int r = rbeg + r0; r < rend; r += rl) {
            const float4 v = src[(size_t)r * c4 + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    s_red[tid] = s;
    __syncthreads();
    if (tid < c4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < rl; ++j) { const float4 v = s_red[j * c4 + tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        partial[(size_t)blockIdx.x * c4 + tid] = t;
    }
}

hipError_t launch_colsum_tall(hipStream_t st, const float* src, int rows, int cols, float alpha, float* dst, float* tmp,
                              size_t tmp_elems)
{
    if (cols % 4 != 0 || cols > 256 || 256 % (cols / 4) != 0 || rows < 8192) return launch_colsum(st, src, rows, cols, cols, alpha, dst);
    int nblk = (int)std::min<size_t>(448, tmp_elems / cols);
    if (nblk < 1) return hipErrorInvalidValue;
    const int rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_tall_partial_kernel, dim3(nblk), dim3(256), 0, st, (const float4*)src, rows, cols / 4, rpb,
                       (float4*)tmp);
    return launch_colsum(st, tmp, nblk, cols, cols, alpha, dst);
}

// =========================================================================================
// generic small SGEMM (row-major): C[M][N] = alpha * op(A)[M][K] * op(B)[K][N] + beta * C, 16x16 LDS tiles.
// Used for the head backward (N = B*K rows; at most a few hundred MFLOP per call).
// =========================================================================================
template <int BK>
__global__ __launch_bounds__(256)
void sgemm_kernel(int ta, int tb, int M, int N, int K, float alpha, const float* __restrict__ A, int lda,
                  const float* __restrict__ B, int ldb, float beta, float* __restrict__ Cm, int ldc)
{
    // These GEMMs are latency-bound (operands are L2-resident, a few hundred blocks): K advances BK = 64 at a time so that every
    // thread has 8 independent loads in flight per barrier pair instead of 2, and 256 at a time (32 loads, four float4 per
    // operand) for the long reductions (K >= 512: 16 -> 4 exposed round trips at K = 1024); the k order of the fma chain is
    // unchanged, so every BK gives the same bits.
    __shared__ float sA[16][BK + 1], sB[BK][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
    float acc = 0.f;
    // operand tiles are fetched as float4 along their contiguous dimension when the leading dimensions allow it (16-byte
    // loads: a quarter of the instructions, full rate of the texture path); scalar loads otherwise / at the edges
    const int t = threadIdx.x;
    const bool vec_ok = !tb && (lda & 3) == 0 && (ldb & 3) == 0 && (K & 3) == 0 && (M & 3) == 0 && (N & 3) == 0 &&
                        (((size_t)A | (size_t)B) & 15) == 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (vec_ok) {
            constexpr int NV = BK / 64;                      // float4 loads per thread and operand
            float4 va[NV], vb4[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                va[q] = vb4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!ta) {                                   // A [M][K]: 16 rows x 64 k per pass, float4 along k
                    const int r = t >> 4, kq = q * 64 + (t & 15) * 4, gr = blockIdx.y * 16 + r;
                    if (gr < M && k0 + kq < K) va[q] = *reinterpret_cast<const float4*>(A + (size_t)gr * lda + k0 + kq);
                } else {                                     // A [K][M]: 64 k x 16 rows per pass, float4 along m
                    const int kk = q * 64 + (t >> 2), gm = blockIdx.y * 16 + (t & 3) * 4;
                    if (k0 + kk < K && gm < M) va[q] = *reinterpret_cast<const float4*>(A + (size_t)(k0 + kk) * lda + gm);
                }
                {                                            // B [K][N]: 64 k x 16 columns per pass, float4 along n
                    const int kk = q * 64 + (t >> 2), gn = blockIdx.x * 16 + (t & 3) * 4;
                    if (k0 + kk < K && gn < N) vb4[q] = *reinterpret_cast<const float4*>(B + (size_t)(k0 + kk) * ldb + gn);
                }
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (!ta) {
                    const int r = t >> 4, kq = q * 64 + (t & 15) * 4;
                    sA[r][kq] = va[q].x; sA[r][kq + 1] = va[q].y; sA[r][kq + 2] = va[q].z; sA[r][kq + 3] = va[q].w;
                } else {
                    const int kk = q * 64 + (t >> 2), mq = (t & 3) * 4;
                    sA[mq][kk] = va[q].x; sA[mq + 1][kk] = va[q].y; sA[mq + 2][kk] = va[q].z; sA[mq + 3][kk] = va[q].w;
                }
                const int kk = q * 64 + (t >> 2), nq = (t & 3) * 4;
                sB[kk][nq] = vb4[q].x; sB[kk][nq + 1] = vb4[q].y; sB[kk][nq + 2] = vb4[q].z; sB[kk][nq + 3] = vb4[q].w;
            }
        } else {
            float ra[BK / 16], rb[BK / 16];
#pragma unroll
            for (int q = 0; q < BK / 16; ++q) {
                const int ka = k0 + q * 16 + tx, kb = k0 + q * 16 + ty;
                ra[q] = (row < M && ka < K) ? (ta ? A[(size_t)ka * lda + row] : A[(size_t)row * lda + ka]) : 0.f;
                rb[q] = (kb < K && col < N) ? (tb ? B[(size_t)col * ldb + kb] : B[(size_t)kb * ldb + col]) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < BK / 16; ++q) { sA[ty][q * 16 + tx] = ra[q]; sB[q * 16 + ty][tx] = rb[q]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK; ++k) acc = fmaf(sA[ty][k], sB[k][tx], acc);
        __syncthreads();
    }
    if (row < M && col < N) {
        float* c = Cm + (size_t)row * ldc + col;
        *c = alpha * acc + (beta != 0.f ? beta * *c : 0.f);
    }
}

// ---- C[M][N] = alpha * A^T . B + beta * C for A [K][M], B [K][N] (the "sum of outer products over rows" shape of every weight gradient of the
// refinement head and of the broadcast layer's latent channels) on v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate.  Both
// operands are K-major, so a lane's operand element of MFMA step k is A[k + lane / 32][m0 + lane % 32]: 128 contiguous bytes per half wave,
// straight from global memory (L2-resident) into the operand register - no LDS, no transposes.  A wave owns a 32 x 32 tile of C;
//   * SPLITK = false: four tiles per block (the 1024 x 512 x 1120 LSTM gradient: 512 tiles; 16 x 16 scalar LDS tiles took 100 us);
//   * SPLITK = true: ONE tile per block, its four waves take a quarter of K each and are added in wave order through LDS (deterministic) -
//     for products with few tiles (64 x 256, K = 1120: 16 tiles).
// MODE 1 writes through the broadcast layer's index map: row = latent channel ci, column = tap * C + co -> gw[co][ci][tap] (ldc = L + 2).
// AROW: A is row-major [M][K] (lda = row stride) instead of K-major - lane i of a half wave reads row m0 + i: 32 cache lines per load
// instruction, 16 MFMA steps per line; for the LSTM gate pre-activations (A = the 224 x 768 head inputs, L1-resident).
template <bool SPLITK, int MODE, bool AROW>
__global__ __launch_bounds__(256)
void sgemm_tn_mfma_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                          float beta, float* __restrict__ Cm, int ldc, int mode_c)
{
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ float s_red[SPLITK ? 3 * 1024 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_n = N / 32, ntiles = (M / 32) * tiles_n;
    const int tile = SPLITK ? (int)blockIdx.x : (int)blockIdx.x * 4 + wv;
    if (tile >= ntiles) return;                              // (wave-uniform; no barrier below unless SPLITK, where it is block-uniform)
    const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
    int k0 = 0, k1 = K;
    if (SPLITK) {
        const int per = ((K + 3) / 4 + 1) & ~1;              // even: a wave's range starts on an MFMA step
        k0 = min(K, wv * per); k1 = min(K, k0 + per);
    }
    const float* pa = AROW ? A + (size_t)(m0 + (lane & 31)) * lda : A + m0 + (lane & 31);
    const float* pb = B + n0 + (lane & 31);
    const int kh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int U = 32;                                    // MFMA steps per group: 2 U loads in flight under the previous group's MFMAs (U = 16:
                                                             // 1024 matrix cycles per group did not cover an L2 round trip under load - 61 us for the LSTM gradient)
    float a[U], b[U];
    auto fetch = [&](int k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + 2 * u + kh;
            const int kc = min(kk, K - 1);                   // unconditional loads at a clamped row; rows past the range count as zero
            const float av = AROW ? pa[kc] : pa[(size_t)kc * lda], bv = pb[(size_t)kc * ldb];
            a[u] = kk < k1 ? av : 0.f; b[u] = bv;
        }
    };
    if (k0 < k1) fetch(k0);
    for (int k = k0; k < k1; k += 2 * U) {
        float ca[U], cb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { ca[u] = a[u]; cb[u] = b[u]; }
        if (k + 2 * U < k1) fetch(k + 2 * U);
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[u], cb[u], acc, 0, 0, 0);
    }
    if (SPLITK) {
        // waves 1 - 3 hand their partial tiles to wave 0: [wave - 1][reg][lane], added in wave order
        if (wv > 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s_red[(wv - 1) * 1024 + i * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += s_red[w * 1024 + i * 64 + lane];
    }
    const int col = n0 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * kh;
        float* c;
        if (MODE == 1) { const int tap = col / mode_c, co = col % mode_c; c = Cm + ((size_t)co * ldc + row) * 9 + tap; }
        else c = Cm + (size_t)row * ldc + col;
        *c = alpha * acc[i] + (beta != 0.f ? beta * *c : 0.f);
    }
}

bool sgemm_tn_mfma_ok(int M, int N, int K) { return M % 32 == 0 && N % 32 == 0 && K >= 1; }

// mode 0: C row-major [M][ldc]; mode 1: the broadcast-layer map (see above), mode_c = C channels, ldc = L + 2
hipError_t launch_sgemm_tn_mfma(hipStream_t st, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                                float beta, float* C, int ldc, int mode, int mode_c, int a_rowmajor)
{
    IOD_XSKIP(2);
    if (!sgemm_tn_mfma_ok(M, N, K) || (mode == 1 && (mode_c < 1 || N != 9 * mode_c))) return hipErrorInvalidValue;
    const int ntiles = (M / 32) * (N / 32);
    // (two waves per SIMD at 512 tiles: one computes while the other waits.  Row-major A = per-slot rows of the refinement head: the summation
    //  order must not depend on how many slots share the launch - a slot's result is the same bits in every batch - so always split)
    const bool splitk = (a_rowmajor || ntiles <= 512) && K >= 64;
#define TN_LAUNCH(SK, MD, AR, GRID) hipLaunchKernelGGL((sgemm_tn_mfma_kernel<SK, MD, AR>), dim3(GRID), dim3(256), 0, st, M, N, K, alpha, A, lda, B, \
                                                       ldb, beta, C, ldc, mode_c)
    if (a_rowmajor) {
        if (mode != 0) return hipErrorInvalidValue;
        if (splitk) TN_LAUNCH(true, 0, true, ntiles); else TN_LAUNCH(false, 0, true, (ntiles + 3) / 4);
    } else if (splitk) { if (mode == 1) TN_LAUNCH(true, 1, false, ntiles); else TN_LAUNCH(true, 0, false, ntiles); }
    else { if (mode == 1) TN_LAUNCH(false, 1, false, (ntiles + 3) / 4); else TN_LAUNCH(false, 0, false, (ntiles + 3) / 4); }
#undef TN_LAUNCH
    return hipGetLastError();
}

hipError_t launch_sgemm(hipStream_t st, int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                        const float* B, int ldb, float beta, float* C, int ldc)
{
    IOD_XSKIP(2);
    if (ta && !tb && sgemm_tn_mfma_ok(M, N, K)) return launch_sgemm_tn_mfma(st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, 0, 0);
    // (the long-K form only where the grid leaves the chip mostly empty: with >= 1024 blocks its 34 KB of LDS costs occupancy -
    // 99 -> 111 us for the 1024 x 512 x 1120 weight gradient)
    if (K >= 512 && ((N + 15) / 16) * ((M + 15) / 16) <= 512)
        hipLaunchKernelGGL(sgemm_kernel<256>, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, st, ta, tb, M, N, K, alpha, A, lda,
                           B, ldb, beta, C, ldc);
    else
        hipLaunchKernelGGL(sgemm_kernel<64>, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, st, ta, tb, M, N, K, alpha, A, lda,
                           B, ldb, beta, C, ldc);
    return hipGetLastError();
}

// =========================================================================================
// spatial-broadcast layer parameter gradients
// =========================================================================================
// RT[n][tap][c] = sum over the border classes in which `tap` stays inside the image of Rc[n][cls][c]
__global__ void l0_tap_sums_kernel(const float* __restrict__ Rc, float* __restrict__ RT, int N, int C)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 9 * C) return;
    const int c = i % C, tap = (i / C) % 9, n = i / (9 * C);
    const int dy = tap / 3, dx = tap % 3;
    float s = 0.f;
    for (int rc = 0; rc < 3; ++rc) {
        if ((rc == 0 && dy == 0) || (rc == 2 && dy == 2)) continue;
        for (int cc = 0; cc < 3; ++cc) {
            if ((cc == 0 && dx == 0) || (cc == 2 && dx == 2)) continue;
            s += Rc[((size_t)n * 9 + rc * 3 + cc) * C + c];
        }
    }
    RT[i] = s;
}

hipError_t launch_l0_tap_sums(hipStream_t st, const float* Rc, float* RT, int N, int C)
{
    hipLaunchKernelGGL(l0_tap_sums_kernel, dim3((N * 9 * C + 255) / 256), dim3(256), 0, st, Rc, RT, N, C);
    return hipGetLastError();
}

// Round 6: the latent-channel weight gradient of the broadcast layer in ONE launch where the fp32-MFMA GEMM does not apply (DIM_LATENT not a
// multiple of 32: the dSprites architectures) - tap sums, z^T . RT and the scatter into gw[co][ci][tap] were three launches of 5 - 7 us each,
// six times per training step.  One thread per (ci, tap, co); the sum over the slot-images runs in four interleaved partial sums (fixed order).
__global__ __launch_bounds__(256)
void l0_latent_wgrad_kernel(const float* __restrict__ Rc, const float* __restrict__ z, int N, int L, int C, float alpha,
                            float* __restrict__ gw)
{
    // block = 32 outputs (consecutive co of one (ci, tap) when C >= 32) x 8 slices of the slot-images; slice s takes n = s, s + 8, ...
    __shared__ float s_part[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    const bool live = i < L * 9 * C;
    const int ii = live ? i : 0;
    const int co = ii % C, tap = (ii / C) % 9, ci = ii / (9 * C);
    const int dy = tap / 3, dx = tap % 3;
    // border classes (row class rc, column class cc) in which the tap stays inside the image: rc in [r0, r1], cc in [c0, c1]
    const int r0 = dy == 0 ? 1 : 0, r1 = dy == 2 ? 1 : 2, c0 = dx == 0 ? 1 : 0, c1 = dx == 2 ? 1 : 2;
    float acc = 0.f;
#pragma unroll 4
    for (int n = sl; n < N; n += 8) {                                   // (unrolled: 36 independent loads in flight instead of 9)
        const float* r = Rc + (size_t)n * 9 * C + co;
        float t = 0.f;
#pragma unroll
        for (int rc = 0; rc < 3; ++rc)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const float v = r[(rc * 3 + cc) * C];                   // (all nine classes are loaded: no divergent addressing, 36 bytes per (n, co))
                t += (rc >= r0 && rc <= r1 && cc >= c0 && cc <= c1) ? v : 0.f;
            }
        acc = fmaf(z[(size_t)n * L + ci], t, acc);
    }
    s_part[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += s_part[q][o];
        gw[((size_t)co * (L + 2) + ci) * 9 + tap] += alpha * t;
    }
}

hipError_t launch_l0_latent_wgrad(hipStream_t st, const float* Rc, const float* z, int N, int L, int C, float alpha, float* gw)
{
    hipLaunchKernelGGL(l0_latent_wgrad_kernel, dim3((L * 9 * C + 31) / 32), dim3(256), 0, st, Rc, z, N, L, C, alpha, gw);
    return hipGetLastError();
}

// gw[co][ci][tap] += alpha * tmp[ci][tap*C + co]  for the latent channels (tmp = z^T . RT)
__global__ void l0_scatter_z_kernel(const float* __restrict__ tmp, int L, int C, float alpha, float* __restrict__ gw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * 9 * C) return;
    const int co = i % C, tap = (i / C) % 9, ci = i / (9 * C);
    gw[((size_t)co * (L + 2) + ci) * 9 + tap] += alpha * tmp[i];
}

hipError_t launch_l0_scatter_z(hipStream_t st, const float* tmp, int L, int C, float alpha, float* gw)
{
    hipLaunchKernelGGL(l0_scatter_z_kernel, dim3((L * 9 * C + 255) / 256), dim3(256), 0, st, tmp, L, C, alpha, gw);
    return hipGetLastError();
}

// coordinate-channel weights and the bias of layer 0 from D[p][c].  Stage 1: block b walks the pixels p = b, b + NB, ...
// with thread = (channel, pixel lane) so that D is read in full rows (the one-block-per-channel form read one float per
// 256-byte row: 151 us for a 4 MB map); 19 sums per channel (9 taps x {x, y} + bias) -> partial[b][19][C].
// Stage 2: fixed-order sum over the NB partials into the gradient accumulators.
constexpr int L0CG_BLOCKS = 512;
__global__ __launch_bounds__(256)
void l0_coord_partial_kernel(const float* __restrict__ D, const float* __restrict__ lin, int S, int C,
                             float* __restrict__ partial)
{
    extern __shared__ float s_cg[];                       // [pixel lanes][19][C]
    const int tid = threadIdx.x, c = tid % C, pl = tid / C, PL = 256 / C;
    float acc[19];
#pragma unroll
    for (int j = 0; j < 19; ++j) acc[j] = 0.f;
    for (int p = blockIdx.x * PL + pl; p < S * S; p += L0CG_BLOCKS * PL) {
        const float v = D[(size_t)p * C + c];
        const int y = p / S, x = p % S;
        acc[18] += v;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
            acc[tap] += lin[xx] * v;
            acc[9 + tap] += lin[yy] * v;
        }
    }
#pragma unroll
    for (int j = 0; j < 19; ++j) s_cg[(pl * 19 + j) * C + c] = acc[j];
    __syncthreads();
    for (int e = tid; e < 19 * C; e += 256) {
        float t = 0.f;
        for (int q = 0; q < PL; ++q) t += s_cg[q * 19 * C + e];
        partial[(size_t)blockIdx.x * 19 * C + e] = t;
    }
}

__global__ void l0_coord_final_kernel(const float* __restrict__ partial, int C, int L, float alpha, float* __restrict__ gw,
                                      float* __restrict__ gb)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 19 * C) return;
    const int j = e / C, co = e % C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int b = 0; b < L0CG_BLOCKS; b += 4) {
        s0 += partial[(size_t)b * 19 * C + e]; s1 += partial[(size_t)(b + 1) * 19 * C + e];
        s2 += partial[(size_t)(b + 2) * 19 * C + e]; s3 += partial[(size_t)(b + 3) * 19 * C + e];
    }
    const float t = alpha * ((s0 + s1) + (s2 + s3));
    if (j < 9) gw[((size_t)co * (L + 2) + L) * 9 + j] += t;
    else if (j < 18) gw[((size_t)co * (L + 2) + L + 1) * 9 + (j - 9)] += t;
    else gb[co] += t;
}

// scratch: at least L0CG_BLOCKS * 19 * C floats
hipError_t launch_l0_coord_grads(hipStream_t st, const float* D, const float* lin, int S, int C, int L, float alpha,
                                 float* gw, float* gb, float* scratch)
{
    if (C > 256 || 256 % C != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(l0_coord_partial_kernel, dim3(L0CG_BLOCKS), dim3(256), (size_t)(256 / C) * 19 * C * sizeof(float), st, D,
                       lin, S, C, scratch);
    hipLaunchKernelGGL(l0_coord_final_kernel, dim3((19 * C + 255) / 256), dim3(256), 0, st, scratch, C, L, alpha, gw, gb);
    return hipGetLastError();
}

// The same gradients from Rsum[y][4][C] = sum over passes and slot-images of the per-row sums {left column, interior columns,
// right column, sum_x lin[x] * d} of d = d(pre-activation 0) (EPI_L0ROWSX epilogue -> l0_reduce_cls_tiles_x -> l0_rowsum_acc):
// no [pixels x C] gradient map is needed.  Per row y with L, M, R, X:
//   bias                          += L + M + R
//   y-channel, tap (ky, kx)       += lin[y + ky - 1] * {M + R, L + M + R, L + M}[kx]             (rows with y + ky - 1 inside)
//   x-channel, tap (ky, kx)       += {(X - lin[0] L) - step (M + R),  X,  (X - lin[S-1] R) + step (L + M)}[kx]
// where lin[x -+ 1] = lin[x] -+ step is used for the shifted weights (step = 2 / (S - 1); the table's own differences deviate
// from it by rounding only, <= 1.2e-7 absolute on weights in [-1, 1]).  One block: thread = (channel, row slice).
__global__ __launch_bounds__(256)
void l0_coord_rows_kernel(const float* __restrict__ Rsum, const float* __restrict__ lin, int S, int C, int L, float alpha,
                          float* __restrict__ gw, float* __restrict__ gb)
{
    __shared__ float s_cg[4 * 19 * 64];
    const int tid = threadIdx.x, c = tid % C, sl = tid / C, NSL = 256 / C;
    const float step = __fdiv_rn(2.f, (float)(S - 1)), lin0 = lin[0], linS = lin[S - 1];
    float acc[19];
#pragma unroll
    for (int j = 0; j < 19; ++j) acc[j] = 0.f;
    for (int y = sl; y < S; y += NSL) {
        const float* r = Rsum + (size_t)y * 4 * C + c;
        const float Lv = r[0], Mv = r[C], Rv = r[2 * C], Xv = r[3 * C];
        const float rs[3] = {Mv + Rv, (Lv + Mv) + Rv, Lv + Mv};
        const float ax[3] = {(Xv - lin0 * Lv) - step * (Mv + Rv), Xv, (Xv - linS * Rv) + step * (Lv + Mv)};
        acc[18] += rs[1];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            if (yy < 0 || yy >= S) continue;
            const float ly = lin[yy];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                acc[ky * 3 + kx] += ax[kx];
                acc[9 + ky * 3 + kx] += ly * rs[kx];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 19; ++j) s_cg[(sl * 19 + j) * C + c] = acc[j];
    __syncthreads();
    for (int e = tid; e < 19 * C; e += 256) {
        float t = 0.f;
        for (int q = 0; q < NSL; ++q) t += s_cg[q * 19 * C + e];
        t *= alpha;
        const int j = e / C, co = e % C;
        if (j < 9) gw[((size_t)co * (L + 2) + L) * 9 + j] += t;
        else if (j < 18) gw[((size_t)co * (L + 2) + L + 1) * 9 + (j - 9)] += t;
        else gb[co] += t;
    }
}

hipError_t launch_l0_coord_grads_rows(hipStream_t st, const float* Rsum, const float* lin, int S, int C, int L, float alpha,
                                      float* gw, float* gb)
{
    if (C > 64 || 256 % C != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(l0_coord_rows_kernel, dim3(1), dim3(256), 0, st, Rsum, lin, S, C, L, alpha, gw, gb);
    return hipGetLastError();
}

// =========================================================================================
// loss and pointwise pieces of the head backward
// =========================================================================================
// loss = -sum_i (i+1)/(T+1) * ELBO_i   (iodine.py:151-158)
__global__ void loss_kernel(const float* __restrict__ scal, int n, float* __restrict__ loss)
{
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += (double)(i + 1) / n * scal[3 * i];
        *loss = (float)(-s);
    }
}

hipError_t launch_loss(hipStream_t st, const float* scal, int n, float* loss)
{
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(64), 0, st, scal, n, loss);
    return hipGetLastError();
}

__global__ void scale_kernel(const float* __restrict__ a, float alpha, float* __restrict__ o, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = alpha * a[i];
}

hipError_t launch_scale(hipStream_t st, const float* a, float alpha, float* o, int n)
{
    IOD_XSKIP(64);
    hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, alpha, o, n);
    return hipGetLastError();
}

__global__ void axpy_kernel(const float* __restrict__ x, float alpha, float* __restrict__ y, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}

// y = (accumulate ? y : 0) + alpha * (*alpha_dev) * x : the hand-over of the accumulated gradients to the caller's flat
// buffer with autograd's incoming grad_output read from device memory (no host round trip, no separate zero-fill)
__global__ void axpy_dev_kernel(const float* __restrict__ x, float alpha, const float* __restrict__ alpha_dev,
                                float* __restrict__ y, int n, int accumulate)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = alpha_dev ? alpha * alpha_dev[0] : alpha;
    y[i] = accumulate ? y[i] + a * x[i] : a * x[i];
}

hipError_t launch_axpy_dev(hipStream_t st, const float* x, float alpha, const float* alpha_dev, float* y, int n, int accumulate)
{
    hipLaunchKernelGGL(axpy_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, alpha, alpha_dev, y, n, accumulate);
    return hipGetLastError();
}

// out[0] = mean(a[0..n)), out[1] = mean(b[0..n)): the two logger scalars of IODINE.forward (iodine.py:156-157)
__global__ void mean2_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, float* __restrict__ out)
{
    const float* src = blockIdx.x == 0 ? a : b;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += (double)src[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(s / n);
}

hipError_t launch_mean2(hipStream_t st, const float* a, const float* b, int n, float* out)
{
    hipLaunchKernelGGL(mean2_kernel, dim3(2), dim3(64), 0, st, a, b, n, out);
    return hipGetLastError();
}

hipError_t launch_axpy(hipStream_t st, const float* x, float alpha, float* y, int n)
{
    IOD_XSKIP(64);
    hipLaunchKernelGGL(axpy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, alpha, y, n);
    return hipGetLastError();
}

// LSTM cell backward, pointwise part.  gates = post-activation (i, f, g, o); c1 = f*c0 + i*g; h1 = o*tanh(c1).
//   in : dc1 (from the update read-out), dh1 / dc1_carry (from the next iteration; may be NULL)
//   out: dgates (pre-activation gradients, [N][4H]), dc0 (carry to the previous iteration)
__global__ void lstm_bwd_pointwise_kernel(const float* __restrict__ gates, const float* __restrict__ c0,
                                          const float* __restrict__ c1, const float* __restrict__ dc1_read,
                                          const float* __restrict__ dh1, const float* __restrict__ dc1_carry,
                                          float* __restrict__ dgates, float* __restrict__ dc0, int N, int H)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * H) return;
    const int n = idx / H, j = idx % H;
    const float* g = gates + (size_t)n * 4 * H;
    const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
    const float tc = tanhf(c1[idx]);
    const float dh = dh1 ? dh1[idx] : 0.f;
    float dc = dc1_read[idx] + (dc1_carry ? dc1_carry[idx] : 0.f);
    dc += dh * go * (1.f - tc * tc);
    float* dg = dgates + (size_t)n * 4 * H;
    dg[j] = dc * gg * gi * (1.f - gi);
    dg[H + j] = dc * c0[idx] * gf * (1.f - gf);
    dg[2 * H + j] = dc * gi * (1.f - gg * gg);
    dg[3 * H + j] = dh * tc * go * (1.f - go);
    dc0[idx] = dc * gf;
}

hipError_t launch_lstm_bwd_pointwise(hipStream_t st, const float* gates, const float* c0, const float* c1,
                                     const float* dc1_read, const float* dh1, const float* dc1_carry, float* dgates,
                                     float* dc0, int N, int H)
{
    IOD_XSKIP(64);
    hipLaunchKernelGGL(lstm_bwd_pointwise_kernel, dim3((N * H + 255) / 256), dim3(256), 0, st, gates, c0, c1, dc1_read,
                       dh1, dc1_carry, dgates, dc0, N, H);
    return hipGetLastError();
}

// u = ELU(ELU(s)): ds = du * ELU'(y) * ELU'(s), y = ELU(s)     (iodine.py:485,565)
__global__ void mlp_bwd_pointwise_kernel(const float* __restrict__ du, int ldu, const float* __restrict__ s,
                                         float* __restrict__ ds, int N, int H)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * H) return;
    const int n = idx / H, j = idx % H;
    const float sv = s[idx];
    const float y = elu1(sv);
    const float g1 = sv > 0.f ? 1.f : y + 1.f;          // ELU'(s)
    const float g2 = y > 0.f ? 1.f : expf(y);           // ELU'(y)
    ds[idx] = du[(size_t)n * ldu + j] * g1 * g2;
}

hipError_t launch_mlp_bwd_pointwise(hipStream_t st, const float* du, int ldu, const float* s, float* ds, int N, int H)
{
    IOD_XSKIP(64);
    hipLaunchKernelGGL(mlp_bwd_pointwise_kernel, dim3((N * H + 255) / 256), dim3(256), 0, st, du, ldu, s, ds, N, H);
    return hipGetLastError();
}

// =========================================================================================
// Back-propagation through time of the refinement HEAD (read-out layers, LSTM cell, double-ELU MLP, average pool) for all T
// iterations in ONE launch.  Every row (slot) of the head is independent of the others - only the weight gradients sum over
// rows, and those stay separate GEMMs over all T * N rows - so a block takes HB rows and walks i = T-1 .. 0 itself:
//   ddm, ddv = alpha_i * d(B ELBO_{i+1}) / d lambda_{i+1}          (lambda_{i+1} = detach(lambda_i) + delta_i, iodine.py:642-643)
//   dc1  = ddm . Wm + ddv . Wv                                      (read-out from the CELL state, iodine.py:488-492)
//   dgates, dc0 = LSTM cell backward (+ the carries dh, dc of iteration i + 1)
//   dh0  = dgates . Whh        (carry)         dxin = dgates . Wih[:, :H]
//   ds   = dxin * ELU'(ELU(s)) * ELU'(s)       dpooled = ds . Wmlp
// It replaces 9 launches per iteration (2 scale, 4 + 1 SGEMM, 2 pointwise: 45 launches, ~0.45 ms per cfg3 step - each of them
// tens of microseconds of latency for a few MFLOP) by one.  The row vectors sit in LDS, a weight element is read once per block and used
// for all HB rows; the k loop is bound by that L2 stream, so the sixteen waves of a block split it and load 16 bytes per lane (head_matvec).
// =========================================================================================
constexpr int HB = 2;      // rows per block
constexpr int HKG = 16;    // k groups = waves of the 1024-thread block; a lane owns four adjacent output columns (16-byte weight loads)

// vout[w][r][j] = sum_k vin[r][k] * W[w][k * ldw[w] + j]   for r < HB, j < J (J, ldw multiples of 4); vin / vout / s_part in LDS.
// The loop streams the weights once per block from L2 and is bound by that stream: wave g sums the k range of group g for the
// column quads lane, lane + 64, ... in ascending k with eight 16-byte loads in flight; the HKG partial sums are added in group order.
template <int NW>
IOD_DEVINL void head_matvec(const float* const (&W)[NW], const int (&ldw)[NW], int K, int J, const float* __restrict__ vin, int ldv,
                            float* const (&vout)[NW], int ldo, float* __restrict__ s_part)
{
    const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int kq = (K + HKG - 1) / HKG, k0 = kg * kq, k1 = min(K, k0 + kq);
    for (int jb = 0; jb < J; jb += 256) {
        const int j = jb + 4 * lane;
        f32x4 acc[NW][HB];
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int r = 0; r < HB; ++r) acc[w][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < J) {
            int k = k0;
            for (; k + 7 < k1; k += 8) {
                f32x4 wv[NW][8];
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int u = 0; u < 8; ++u) wv[w][u] = *reinterpret_cast<const f32x4*>(W[w] + (size_t)(k + u) * ldw[w] + j);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int r = 0; r < HB; ++r) {
                        const float x = vin[r * ldv + k + u];
#pragma unroll
                        for (int w = 0; w < NW; ++w) acc[w][r] += x * wv[w][u];
                    }
            }
            for (; k < k1; ++k)
#pragma unroll
                for (int r = 0; r < HB; ++r) {
                    const float x = vin[r * ldv + k];
#pragma unroll
                    for (int w = 0; w < NW; ++w) acc[w][r] += x * *reinterpret_cast<const f32x4*>(W[w] + (size_t)k * ldw[w] + j);
                }
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int r = 0; r < HB; ++r) *reinterpret_cast<f32x4*>(s_part + ((kg * NW + w) * HB + r) * 256 + 4 * lane) = acc[w][r];
        }
        __syncthreads();
        // fixed-order sum over the k groups
        const int ncol = min(256, J - jb);
        for (int o = threadIdx.x; o < NW * HB * ncol; o += blockDim.x) {
            const int c = o % ncol, wr = o / ncol;
            float sum = s_part[(0 * NW * HB + wr) * 256 + c];
#pragma unroll
            for (int g = 1; g < HKG; ++g) sum += s_part[(g * NW * HB + wr) * 256 + c];
            vout[wr / HB][(wr % HB) * ldo + jb + c] = sum;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024)
void head_bptt_kernel(const float* __restrict__ g_pm, const float* __restrict__ g_plv, const float* __restrict__ gates,
                      const float* __restrict__ cst, const float* __restrict__ u, const float* __restrict__ Wm,
                      const float* __restrict__ Wv, const float* __restrict__ Whh, const float* __restrict__ Wih,
                      const float* __restrict__ Wmlp, float* __restrict__ ddm_o, float* __restrict__ ddv_o,
                      float* __restrict__ dgates_o, float* __restrict__ ds_o, float* __restrict__ dpooled_o, int T, int N, int B,
                      int L, int H, int Cr)
{
    extern __shared__ __attribute__((aligned(16))) float s_hb[];
    float* s_dd = s_hb;                              // [HB][2L]   ddm | ddv
    float* s_dc1 = s_dd + HB * 2 * L;                // [HB][H]    (two halves summed: see below)
    float* s_dc1b = s_dc1 + HB * H;                  // [HB][H]
    float* s_dg = s_dc1b + HB * H;                   // [HB][4H]
    float* s_dh = s_dg + HB * 4 * H;                 // [HB][H]    carry dh (from iteration i + 1)
    float* s_dcc = s_dh + HB * H;                    // [HB][H]    carry dc
    float* s_dx = s_dcc + HB * H;                    // [HB][H]    dxin, then ds
    float* s_dhn = s_dx + HB * H;                    // [HB][H]    new carry dh
    float* s_part = s_dhn + HB * H;                  // [HKG][2][HB][256] partial sums of head_matvec (16-byte aligned: all sizes are multiples of 4)
    const int tid = threadIdx.x, n0 = blockIdx.x * HB;
    const int IN = H + 4 * L;
    for (int idx = tid; idx < HB * H; idx += 1024) { s_dh[idx] = 0.f; s_dcc[idx] = 0.f; }
    __syncthreads();
    for (int i = T - 1; i >= 0; --i) {
        const float alpha = -((float)(i + 2) / (float)(T + 1)) / (float)B;
        // 1. scaled posterior gradients of iteration i + 1
        for (int idx = tid; idx < HB * L; idx += 1024) {
            const int r = idx / L, l = idx % L, n = min(n0 + r, N - 1);
            const float a = alpha * g_pm[((size_t)(i + 1) * N + n) * L + l], b = alpha * g_plv[((size_t)(i + 1) * N + n) * L + l];
            s_dd[r * 2 * L + l] = a; s_dd[r * 2 * L + L + l] = b;
            if (n0 + r < N) { ddm_o[((size_t)i * N + n) * L + l] = a; ddv_o[((size_t)i * N + n) * L + l] = b; }
        }
        __syncthreads();
        // 2. read-out layers: dc1 = ddm . Wm (+) ddv . Wv, the two products summed in that order like the two SGEMM calls it replaces
        {
            const float* const W0[1] = {Wm}; const float* const W1[1] = {Wv};
            const int ld0[1] = {H};
            float* const o0[1] = {s_dc1}; float* const o1[1] = {s_dc1b};
            head_matvec<1>(W0, ld0, L, H, s_dd, 2 * L, o0, H, s_part);
            head_matvec<1>(W1, ld0, L, H, s_dd + L, 2 * L, o1, H, s_part);
        }
        __syncthreads();
        // 3. LSTM cell backward (pointwise)
        for (int idx = tid; idx < HB * H; idx += 1024) {
            const int r = idx / H, j = idx % H, n = min(n0 + r, N - 1);
            const float* g = gates + ((size_t)i * N + n) * 4 * H;
            const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
            const float c0v = cst[((size_t)i * N + n) * H + j], tc = tanhf(cst[((size_t)(i + 1) * N + n) * H + j]);
            const float dh = s_dh[idx];
            float dc = (s_dc1[idx] + s_dc1b[idx]) + s_dcc[idx];
            dc += dh * go * (1.f - tc * tc);
            const float d0 = dc * gg * gi * (1.f - gi), d1 = dc * c0v * gf * (1.f - gf), d2 = dc * gi * (1.f - gg * gg),
                        d3 = dh * tc * go * (1.f - go);
            float* sg = s_dg + r * 4 * H;
            sg[j] = d0; sg[H + j] = d1; sg[2 * H + j] = d2; sg[3 * H + j] = d3;
            if (n0 + r < N) {
                float* og = dgates_o + ((size_t)i * N + n) * 4 * H;
                og[j] = d0; og[H + j] = d1; og[2 * H + j] = d2; og[3 * H + j] = d3;
            }
            s_dcc[idx] = dc * gf;                                           // carry dc for iteration i - 1 (own element: no hazard)
        }
        __syncthreads();
        // 4. + 5. dh0 = dgates . Whh (carry), dxin = dgates . Wih[:, :H]: one pass over k for both
        {
            const float* const W[2] = {Whh, Wih};
            const int ld[2] = {H, IN};
            float* const out[2] = {s_dhn, s_dx};
            head_matvec<2>(W, ld, 4 * H, H, s_dg, 4 * H, out, H, s_part);
        }
        __syncthreads();
        // 6. double-ELU MLP backward (pointwise); the new dh carry moves into place
        for (int idx = tid; idx < HB * H; idx += 1024) {
            const int r = idx / H, j = idx % H, n = min(n0 + r, N - 1);
            const float sv = u[((size_t)i * N + n) * H + j];
            const float y = elu1(sv);
            const float g1 = sv > 0.f ? 1.f : y + 1.f, g2 = y > 0.f ? 1.f : expf(y);
            const float d = s_dx[idx] * g1 * g2;
            s_dx[idx] = d;
            if (n0 + r < N) ds_o[((size_t)i * N + n) * H + j] = d;
            s_dh[idx] = s_dhn[idx];
        }
        __syncthreads();
        // 7. average-pool input gradient: dpooled = ds . Wmlp  (written straight to global: s_dc1 is free, used as staging)
        {
            const float* const W[1] = {Wmlp};
            const int ld[1] = {Cr};
            float* const out[1] = {s_dc1};
            head_matvec<1>(W, ld, H, Cr, s_dx, H, out, H, s_part);
        }
        __syncthreads();
        for (int idx = tid; idx < HB * Cr; idx += 1024) {
            const int r = idx / Cr, c = idx % Cr;
            if (n0 + r < N) dpooled_o[((size_t)i * N + n0 + r) * Cr + c] = s_dc1[r * H + c];
        }
        __syncthreads();
    }
}

// does the fused kernel's LDS footprint fit (it does for every shipped configuration; MLP_UNITS >= 512 falls back to the launch sequence)
static size_t head_bptt_lds(int L, int H) { return ((size_t)HB * (2 * L + 10 * H) + (size_t)HKG * 2 * HB * 256) * sizeof(float); }
// (head_matvec reads the weights as 16-byte f32x4 rows: every row length / leading dimension - H, H + 4L, Cr, L - must be a multiple of 4)
bool head_bptt_fits(int L, int H, int Cr) { return Cr <= H && H % 4 == 0 && Cr % 4 == 0 && L % 4 == 0 && head_bptt_lds(L, H) <= 160 * 1024; }

hipError_t launch_head_bptt(hipStream_t st, const float* g_pm, const float* g_plv, const float* gates, const float* cst, const float* u,
                            const float* Wm, const float* Wv, const float* Whh, const float* Wih, const float* Wmlp, float* ddm,
                            float* ddv, float* dgates, float* ds, float* dpooled, int T, int N, int B, int L, int H, int Cr)
{
    IOD_XSKIP(2);
    if (Cr > H) return hipErrorInvalidValue;                                // (the pool gradient is staged in an [HB][H] buffer)
    const size_t lds = head_bptt_lds(L, H);
    static std::atomic<unsigned> attr_devs{0};
    if (hipError_t e = iod_set_max_lds((const void*)head_bptt_kernel, 160 * 1024, attr_devs); e != hipSuccess) return e;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_bptt_kernel, dim3((N + HB - 1) / HB), dim3(64 * HKG), lds, st, g_pm, g_plv, gates, cst, u, Wm, Wv, Whh, Wih, Wmlp,
                       ddm, ddv, dgates, ds, dpooled, T, N, B, L, H, Cr);
    return hipGetLastError();
}

// avg-pool backward fused with the ELU derivative of the last refinement conv layer
__global__ void pool_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ act, float* __restrict__ dpre,
                                int PL, int C, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % C;
    const size_t n = i / ((size_t)PL * C);
    dpre[i] = dpooled[n * C + c] / (float)PL * elu1_grad_from_out(act[i]);
}

hipError_t launch_pool_bwd(hipStream_t st, const float* dpooled, const float* act, float* dpre, int N, int PL, int C)
{
    IOD_XSKIP(64);
    const size_t total = (size_t)N * PL * C;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dpooled, act, dpre, PL, C,
                       total);
    return hipGetLastError();
}

// =========================================================================================
// Weight gradient of the decoder OUTPUT conv (C -> 4, lib/modeling/iodine.py:422) in GEMM form, split-fp16.
//   dW[co][ci][tap] = sum_q a[q][ci] * g[q - off(tap)][co]        (q = p + off(tap): the shift is moved onto g)
// is ONE [36 x pixels] . [pixels x C] product per tile with rows j = tap*4 + co (36 of 64 used) instead of nine
// products with N = 4 padded to 32: 4.5x fewer MFMAs, and the 64-channel operand needs no halo (the 4-channel g does).
//   A operand (rows j): lane (j, kh) reads its 8 K-consecutive pixels from the fp16 plane of channel co = j & 3 at the
//     row / column offset of tap = j >> 2 (5 dwords + v_alignbit with a per-lane shift)
//   B operand (cols ci): transposed fp16 planes of the activation tile, aligned b128 reads
//   tile = 4 x 16 pixels, K = 16 per MFMA = one tile row; persistent blocks, fixed-order partial tiles [tap][ci][4].
// =========================================================================================
template <int C>
__global__ __launch_bounds__(256, 4)
void dec_out_wgrad_gemm_f16x3_kernel(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ part,
                                     float* __restrict__ part_b, int S, int ntiles, int tiles_x, int tiles_y)
{
    constexpr float alpha = 1.f; constexpr int accum = 0;    // (this two-kernel form reduces after every launch: option out_bwd_fused 0)
    constexpr int NTT = C / 32;                          // ci tiles
    constexpr int KS = 2 / NTT;                          // waves = 2 (j tiles) x NTT x KS
    constexpr int TH = 4, RW = TH / KS;
    constexpr int BPL = TH * 8 + 4;                      // dwords per activation channel plane (BPL/4 odd)
    constexpr int GROW = 12, GPL = (TH + 2) * GROW;      // g planes: 6 halo rows x (18 columns = 9 dwords, padded to 12)
    constexpr int A4 = C / 4;
    constexpr int NB_UNITS = TH * 8 * A4, NBU = NB_UNITS / 256;
    constexpr int NG_UNITS = (TH + 2) * 9;
    static_assert(NB_UNITS % 256 == 0, "activation tile units");

    __shared__ __attribute__((aligned(16))) unsigned s_b[2 * C * BPL];
    __shared__ __attribute__((aligned(16))) unsigned s_g[2 * 4 * GPL];
    __shared__ float s_max[8];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int mj = wv & 1, ni = (wv >> 1) % NTT, ks = wv / (2 * NTT);
    const int j = mj * 32 + li, tap = j >> 2, co = j & 3;
    const bool j_ok = tap < 9;
    const int ky = j_ok ? tap / 3 : 1, kx = j_ok ? tap % 3 : 1;
    const int h0 = 8 * kh + 2 - kx;                      // first half-word (column + 1) of this lane's 8 pixels
    const unsigned sh = (h0 & 1) * 16;
    const int g_base = co * GPL + (2 - ky) * GROW + (h0 >> 1);
    const int ci = ni * 32 + li;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float sa = 1.f, sd = 1.f, acc_prod = 1.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const float* a_n = a + (size_t)n * S * S * C;
        const float4* g_n = reinterpret_cast<const float4*>(g) + (size_t)n * S * S;

        float4 rb[NBU][2], rg[2];
        float ma = 0.f, md = 0.f;
#pragma unroll
        for (int k = 0; k < NBU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % A4, tt = u / A4, p = tt % 8, row = tt / 8;
            const float* src = a_n + ((size_t)(ty * TH + row) * S + tx * 16 + 2 * p) * C + c4 * 4;
            rb[k][0] = *reinterpret_cast<const float4*>(src);
            rb[k][1] = *reinterpret_cast<const float4*>(src + C);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                ma = fmaxf(ma, fmaxf(fmaxf(fabsf(rb[k][q].x), fabsf(rb[k][q].y)), fmaxf(fabsf(rb[k][q].z), fabsf(rb[k][q].w))));
        }
        rg[0] = rg[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < NG_UNITS) {                              // g halo tile: pixel pairs (columns 2p-1, 2p), one float4 = 4 channels
            const int p = tid % 9, row = tid / 9;
            const int gy = ty * TH - 1 + row, gx = tx * 16 - 1 + 2 * p;
            if (gy >= 0 && gy < S) {
                if (gx >= 0 && gx < S) rg[0] = g_n[(size_t)gy * S + gx];
                if (gx + 1 < S) rg[1] = g_n[(size_t)gy * S + gx + 1];
            }
            const bool rin = row >= 1 && row <= TH;        // bias gradient: interior pixels only (each pixel once per launch)
            if (rin && p >= 1) { bsum.x += rg[0].x; bsum.y += rg[0].y; bsum.z += rg[0].z; bsum.w += rg[0].w; }
            if (rin && p <= 7) { bsum.x += rg[1].x; bsum.y += rg[1].y; bsum.z += rg[1].z; bsum.w += rg[1].w; }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                md = fmaxf(md, fmaxf(fmaxf(fabsf(rg[q].x), fabsf(rg[q].y)), fmaxf(fabsf(rg[q].z), fabsf(rg[q].w))));
        }
        ma = wave_max_f32(ma);
        md = wave_max_f32(md);
        if (lane == 0) { s_max[wv] = ma; s_max[4 + wv] = md; }
        __syncthreads();                                   // every wave is also done with the previous tile's planes
        ma = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        md = fmaxf(fmaxf(s_max[4], s_max[5]), fmaxf(s_max[6], s_max[7]));
        sa = tile_scale(ma, sa);
        sd = tile_scale(md, sd);
        const float prod = sa * sd;
        if (prod != acc_prod) {                            // block-uniform; exact (powers of two)
            const float r = prod / acc_prod;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] *= r;
            acc_prod = prod;
        }
#pragma unroll
        for (int k = 0; k < NBU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % A4, tt = u / A4, p = tt % 8, row = tt / 8;
            const int rot = c4 & 3;
            const float4 q0 = rot4(rb[k][0], rot), q1 = rot4(rb[k][1], rot);
            const float x0[4] = {q0.x, q0.y, q0.z, q0.w}, x1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned lo;
                const unsigned hi = pack_hi_lo(x0[e] * sa, x1[e] * sa, lo);
                const int ch = c4 * 4 + ((e + rot) & 3);
                s_b[(0 * C + ch) * BPL + row * 8 + p] = hi;
                s_b[(1 * C + ch) * BPL + row * 8 + p] = lo;
            }
        }
        if (tid < NG_UNITS) {
            const int p = tid % 9, row = tid / 9;
            const float x0[4] = {rg[0].x, rg[0].y, rg[0].z, rg[0].w}, x1[4] = {rg[1].x, rg[1].y, rg[1].z, rg[1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned lo;
                const unsigned hi = pack_hi_lo(x0[e] * sd, x1[e] * sd, lo);
                s_g[(0 * 4 + e) * GPL + row * GROW + p] = hi;
                s_g[(1 * 4 + e) * GPL + row * GROW + p] = lo;
            }
        }
        __syncthreads();

#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = ks * RW + rr;
            h16x8 A[2], B[2];
#pragma unroll
            for (int term = 0; term < 2; ++term) {
                const unsigned* pg = s_g + term * 4 * GPL + g_base + r * GROW;
                const unsigned v0 = pg[0], v1 = pg[1], v2 = pg[2], v3 = pg[3], v4 = pg[4];
                uint4 m;
                m.x = __builtin_amdgcn_alignbit(v1, v0, sh);
                m.y = __builtin_amdgcn_alignbit(v2, v1, sh);
                m.z = __builtin_amdgcn_alignbit(v3, v2, sh);
                m.w = __builtin_amdgcn_alignbit(v4, v3, sh);
                if (!j_ok) m = make_uint4(0u, 0u, 0u, 0u);
                __builtin_memcpy(&A[term], &m, 16);
                const uint4 vb = *reinterpret_cast<const uint4*>(s_b + (term * C + ci) * BPL + r * 8 + 4 * kh);
                __builtin_memcpy(&B[term], &vb, 16);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], B[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[0], acc, 0, 0, 0);
        }
    }

    // rows of the accumulator are j = tap*4 + co: registers 4q .. 4q+3 of a lane are the 4 output channels of one tap
    // (accum / alpha: the block's partial tile accumulates over the decoder passes of a training step, see conv3x3_wgrad_f16x3_ws_kernel)
    const float inv = alpha / acc_prod;
    float* pw = part + (size_t)(blockIdx.x * KS + ks) * 9 * C * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tp = mj * 8 + 2 * q + kh;
        if (tp < 9) {
            float4* dst4 = reinterpret_cast<float4*>(pw + ((size_t)tp * C + ci) * 4);
            float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (accum) o4 = *dst4;
            *dst4 = make_float4(o4.x + acc[4 * q] * inv, o4.y + acc[4 * q + 1] * inv, o4.z + acc[4 * q + 2] * inv, o4.w + acc[4 * q + 3] * inv);
        }
    }
    __syncthreads();
    float4* s_red = reinterpret_cast<float4*>(s_b);
    s_red[tid] = bsum;
    __syncthreads();
    if (tid == 0) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < 256; ++q) { const float4 v = s_red[q]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
        float4* pb = reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * 4);
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (accum) o4 = *pb;
        *pb = make_float4(o4.x + alpha * t4.x, o4.y + alpha * t4.y, o4.z + alpha * t4.z, o4.w + alpha * t4.w);
    }
}

// part: nparts x [9][c][4], part_b: nbias_parts x [4]
hipError_t launch_dec_out_wgrad_gemm_f16x3(hipStream_t st, const float* a, const float* g, float* part, float* part_b, int N,
                                           int S, int c, int* nparts, int* nbias_parts)
{
    IOD_XSKIP(2048);
    if (S % 16 != 0 || (c != 64 && c != 32)) return hipErrorInvalidValue;
    const int tiles_x = S / 16, tiles_y = S / 4, ntiles = N * tiles_x * tiles_y;
    const int blocks = ntiles < 1024 ? ntiles : 1024;
    if (c == 64)
        hipLaunchKernelGGL((dec_out_wgrad_gemm_f16x3_kernel<64>), dim3(blocks), dim3(256), 0, st, a, g, part, part_b, S, ntiles,
                           tiles_x, tiles_y);
    else
        hipLaunchKernelGGL((dec_out_wgrad_gemm_f16x3_kernel<32>), dim3(blocks), dim3(256), 0, st, a, g, part, part_b, S, ntiles,
                           tiles_x, tiles_y);
    *nparts = blocks * (c == 64 ? 1 : 2);
    *nbias_parts = blocks;
    return hipGetLastError();
}

// =========================================================================================
// Output conv backward in ONE pass over the saved activation (training): the data gradient of dec_out_dgrad_f16x3_kernel
// (kernels_out.hip; 4 -> C channels, times ELU') and the weight / bias gradient of dec_out_wgrad_gemm_f16x3_kernel above
// both stream the same 0.94 GB tensor (cfg3) and the same 4-channel gradient; run back to back they are 0.35 + 0.23 ms per
// decoder pass, both HBM-bound.  Here a persistent block takes 4 x 16-pixel tiles: the activation tile is fetched once
// (whole pixels, kept in registers for the ELU' factor), split into the transposed fp16 planes of the weight-gradient GEMM
// (K = pixels), the gradient halo is staged twice (fp16 planes for the weight gradient, fp32 for the data gradient's
// [pixels x 36] operand), and after the MFMAs of both products the data-gradient tile is transposed through the plane buffer
// and leaves as whole pixels.  The next tile's loads are issued before the MFMAs.  Same arithmetic per product as the two
// kernels it replaces (same packs, same three passes; the data gradient's power-of-two scale is a function of the tile's own
// halo only, so its result does not depend on which block ran the tile).
// =========================================================================================
template <int C, bool PF>
__global__ __launch_bounds__(256, (C == 64 && PF) ? 3 : 4)
void dec_out_bwd_fused_f16x3_kernel(const float* __restrict__ a, const float* __restrict__ g, const uint4* __restrict__ wpk,
                                    const float* __restrict__ wmeta, float* __restrict__ out, float* __restrict__ tmax,
                                    float* __restrict__ part, float* __restrict__ part_b, int S, int ntiles, int tiles_x,
                                    int tiles_y, float alpha, int accum)
{
    constexpr int NTT = C / 32;                          // ci tiles
    constexpr int KS = 2 / NTT;                          // waves = 2 (j tiles) x NTT x KS
    constexpr int TH = 4, RW = TH / KS;
    constexpr int BPL = TH * 8 + 4;                      // dwords per activation channel plane (BPL/4 odd)
    constexpr int GROW = 12, GPL = (TH + 2) * GROW;      // g planes: 6 halo rows x (18 columns = 9 dwords, padded to 12)
    constexpr int A4 = C / 4;
    constexpr int NB_UNITS = TH * 8 * A4, NBU = NB_UNITS / 256;
    constexpr int NG_UNITS = (TH + 2) * 9;
    constexpr int HW = 18;                               // fp32 gradient halo: 6 rows x 18 columns
    constexpr int W_U4 = 3 * 2 * 2 * C;                  // packed data-gradient weights (pack_dec_out_dgrad_kernel)
    constexpr int EPD = C + 4;                           // dwords per transposed pixel
    constexpr int NDW = 2 * NTT;                         // waves that own a 32 channel x 32 pixel block of the data gradient
    static_assert(NB_UNITS % 256 == 0, "activation tile units");
    static_assert(2 * C * BPL >= TH * 16 * EPD, "transposition region fits the plane buffer");

    __shared__ __attribute__((aligned(16))) unsigned s_b[2 * C * BPL];   // fp16 planes, then the transposed output tile
    __shared__ __attribute__((aligned(16))) unsigned s_g[2 * 4 * GPL];
    __shared__ __attribute__((aligned(16))) float4 s_g32[(TH + 2) * HW];
    __shared__ __attribute__((aligned(16))) uint4 s_w[W_U4];
    __shared__ float s_max[8];
    __shared__ float s_omax[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int mj = wv & 1, ni = (wv >> 1) % NTT, ks = wv / (2 * NTT);
    const int j = mj * 32 + li, tap = j >> 2, co = j & 3;
    const bool j_ok = tap < 9;
    const int ky = j_ok ? tap / 3 : 1, kx = j_ok ? tap % 3 : 1;
    const int h0 = 8 * kh + 2 - kx;                      // first half-word (column + 1) of this lane's 8 pixels
    const unsigned sh = (h0 & 1) * 16;
    const int g_base = co * GPL + (2 - ky) * GROW + (h0 >> 1);
    const int ci = ni * 32 + li;
    // data gradient: wave -> (channel half dch, pixel half dph); lane li = pixel of the half (row 2*dph + li/16, column li%16)
    const bool d_on = wv < NDW;
    const int dch = wv % NTT, dph = (wv / NTT) & 1;
    const int dpy = 2 * dph + (li >> 4), dpx = li & 15;

    for (int idx = tid; idx < W_U4; idx += 256) s_w[idx] = wpk[idx];
    const float winv = wmeta[1];

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float sa = 1.f, sd = 1.f, acc_prod = 1.f;

    float4 rb[NBU][2], rg[2];
    auto fetch = [&](int tile, float4 (&fb)[NBU][2], float4 (&fg)[2]) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const float* a_n = a + (size_t)n * S * S * C;
        const float4* g_n = reinterpret_cast<const float4*>(g) + (size_t)n * S * S;
#pragma unroll
        for (int k = 0; k < NBU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % A4, tt = u / A4, p = tt % 8, row = tt / 8;
            const float* src = a_n + ((size_t)(ty * TH + row) * S + tx * 16 + 2 * p) * C + c4 * 4;
            fb[k][0] = *reinterpret_cast<const float4*>(src);
            fb[k][1] = *reinterpret_cast<const float4*>(src + C);
        }
        fg[0] = fg[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < NG_UNITS) {                              // g halo tile: pixel pairs (columns 2p-1, 2p), one float4 = 4 channels
            const int p = tid % 9, row = tid / 9;
            const int gy = ty * TH - 1 + row, gx = tx * 16 - 1 + 2 * p;
            if (gy >= 0 && gy < S) {
                if (gx >= 0 && gx < S) fg[0] = g_n[(size_t)gy * S + gx];
                if (gx + 1 < S) fg[1] = g_n[(size_t)gy * S + gx + 1];
            }
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x, rb, rg);
    int prev_tm = -1;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        float* out_n = out + (size_t)n * S * S * C;

        float ma = 0.f, md = 0.f;
#pragma unroll
        for (int k = 0; k < NBU; ++k)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                ma = fmaxf(ma, fmaxf(fmaxf(fabsf(rb[k][q].x), fabsf(rb[k][q].y)), fmaxf(fabsf(rb[k][q].z), fabsf(rb[k][q].w))));
        if (tid < NG_UNITS) {
            const int p = tid % 9, row = tid / 9;
            const bool rin = row >= 1 && row <= TH;        // bias gradient: interior pixels only (each pixel once per launch)
            if (rin && p >= 1) { bsum.x += rg[0].x; bsum.y += rg[0].y; bsum.z += rg[0].z; bsum.w += rg[0].w; }
            if (rin && p <= 7) { bsum.x += rg[1].x; bsum.y += rg[1].y; bsum.z += rg[1].z; bsum.w += rg[1].w; }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                md = fmaxf(md, fmaxf(fmaxf(fabsf(rg[q].x), fabsf(rg[q].y)), fmaxf(fabsf(rg[q].z), fabsf(rg[q].w))));
        }
        ma = wave_max_f32(ma);
        md = wave_max_f32(md);
        if (lane == 0) { s_max[wv] = ma; s_max[4 + wv] = md; }
        __syncthreads();                                   // (A) every wave is also done with the previous tile's LDS
        if (tmax && prev_tm >= 0 && tid < 2)               // side buffer of the previous tile (its wave maxima are visible now)
            tmax[prev_tm + tid] = fmaxf(fmaxf(s_omax[0], s_omax[1]), fmaxf(s_omax[2], s_omax[3]));
        ma = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        md = fmaxf(fmaxf(s_max[4], s_max[5]), fmaxf(s_max[6], s_max[7]));
        sa = tile_scale(ma, sa);
        sd = tile_scale(md, sd);
        const float sdg = tile_scale(md, 1.f);             // data gradient: a function of this tile's halo only
        const float prod = sa * sd;
        if (prod != acc_prod) {                            // block-uniform; exact (powers of two)
            const float r = prod / acc_prod;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] *= r;
            acc_prod = prod;
        }
#pragma unroll
        for (int k = 0; k < NBU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % A4, tt = u / A4, p = tt % 8, row = tt / 8;
            const int rot = c4 & 3;
            const float4 q0 = rot4(rb[k][0], rot), q1 = rot4(rb[k][1], rot);
            const float x0[4] = {q0.x, q0.y, q0.z, q0.w}, x1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned lo;
                const unsigned hi = pack_hi_lo(x0[e] * sa, x1[e] * sa, lo);
                const int ch = c4 * 4 + ((e + rot) & 3);
                s_b[(0 * C + ch) * BPL + row * 8 + p] = hi;
                s_b[(1 * C + ch) * BPL + row * 8 + p] = lo;
            }
        }
        if (tid < NG_UNITS) {
            const int p = tid % 9, row = tid / 9;
            const float x0[4] = {rg[0].x, rg[0].y, rg[0].z, rg[0].w}, x1[4] = {rg[1].x, rg[1].y, rg[1].z, rg[1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned lo;
                const unsigned hi = pack_hi_lo(x0[e] * sd, x1[e] * sd, lo);
                s_g[(0 * 4 + e) * GPL + row * GROW + p] = hi;
                s_g[(1 * 4 + e) * GPL + row * GROW + p] = lo;
            }
            s_g32[row * HW + 2 * p] = rg[0];
            s_g32[row * HW + 2 * p + 1] = rg[1];
        }
        __syncthreads();                                   // (B)

        // next tile's loads fly under the MFMAs and the epilogue (rg is free; the activation goes to a second set)
        float4 rbn[NBU][2], rgn[2];
        const bool has_next = tile + (int)gridDim.x < ntiles;
        if (PF && has_next) fetch(tile + gridDim.x, rbn, rgn);

        // ---- weight gradient: rows j = (tap, co), columns ci, K = the pixels of one tile row per MFMA ----
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = ks * RW + rr;
            h16x8 A[2], B[2];
#pragma unroll
            for (int term = 0; term < 2; ++term) {
                const unsigned* pg = s_g + term * 4 * GPL + g_base + r * GROW;
                const unsigned v0 = pg[0], v1 = pg[1], v2 = pg[2], v3 = pg[3], v4 = pg[4];
                uint4 m;
                m.x = __builtin_amdgcn_alignbit(v1, v0, sh);
                m.y = __builtin_amdgcn_alignbit(v2, v1, sh);
                m.z = __builtin_amdgcn_alignbit(v3, v2, sh);
                m.w = __builtin_amdgcn_alignbit(v4, v3, sh);
                if (!j_ok) m = make_uint4(0u, 0u, 0u, 0u);
                __builtin_memcpy(&A[term], &m, 16);
                const uint4 vb = *reinterpret_cast<const uint4*>(s_b + (term * C + ci) * BPL + r * 8 + 4 * kh);
                __builtin_memcpy(&B[term], &vb, 16);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], B[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[0], acc, 0, 0, 0);
        }
        // ---- data gradient: rows = channels (weights first), columns = 32 pixels, K = (tap, co) padded to 48 ----
        f32x16 dacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
        if (d_on) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int t0 = 4 * c + 2 * kh;                      // this lane's two taps of the chunk
                float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
                if (t0 < 9) q0 = s_g32[(dpy + t0 / 3) * HW + dpx + t0 % 3];
                if (t0 + 1 < 9) q1 = s_g32[(dpy + (t0 + 1) / 3) * HW + dpx + (t0 + 1) % 3];
                unsigned l0, l1, l2, l3;
                const unsigned u0 = pack_hi_lo(q0.x * sdg, q0.y * sdg, l0), u1 = pack_hi_lo(q0.z * sdg, q0.w * sdg, l1);
                const unsigned u2 = pack_hi_lo(q1.x * sdg, q1.y * sdg, l2), u3 = pack_hi_lo(q1.z * sdg, q1.w * sdg, l3);
                const uint4 uh = make_uint4(u0, u1, u2, u3), ul = make_uint4(l0, l1, l2, l3);
                h16x8 ah, al, bh, bl;
                __builtin_memcpy(&ah, &uh, 16); __builtin_memcpy(&al, &ul, 16);
                const uint4 wh = s_w[((c * 2 + 0) * 2 + kh) * C + dch * 32 + li];
                const uint4 wl = s_w[((c * 2 + 1) * 2 + kh) * C + dch * 32 + li];
                __builtin_memcpy(&bh, &wh, 16); __builtin_memcpy(&bl, &wl, 16);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, dacc, 0, 0, 0);
            }
        }
        __syncthreads();                                   // (C) the planes are dead: the buffer becomes the transposed tile
        float* ep = reinterpret_cast<float*>(s_b);
        if (d_on) {
            const float inv = winv / sdg;
            // lane (li, kh) holds channels dch*32 + 8*g4 + 4*kh .. +3 of pixel dph*32 + li
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<f32x4*>(ep + (dph * 32 + li) * EPD + dch * 32 + 8 * g4 + 4 * kh) =
                    f32x4{dacc[4 * g4] * inv, dacc[4 * g4 + 1] * inv, dacc[4 * g4 + 2] * inv, dacc[4 * g4 + 3] * inv};
        }
        __syncthreads();                                   // (D)
        float omax = 0.f;
#pragma unroll
        for (int k = 0; k < NBU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % A4, tt = u / A4, p = tt % 8, row = tt / 8;
            float* dst = out_n + ((size_t)(ty * TH + row) * S + tx * 16 + 2 * p) * C + c4 * 4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>(ep + (row * 16 + 2 * p + q) * EPD + c4 * 4);
                const float4 a4 = rb[k][q];
                v.x *= elu1_grad_from_out(a4.x); v.y *= elu1_grad_from_out(a4.y);
                v.z *= elu1_grad_from_out(a4.z); v.w *= elu1_grad_from_out(a4.w);
                omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                *reinterpret_cast<f32x4*>(dst + q * C) = v;
            }
        }
        omax = wave_max_f32(omax);
        if (lane == 0) s_omax[wv] = omax;
        // 8 x 16 cells of the side buffer: this 4-row tile owns two of its cell's four slots
        prev_tm = (((n * (S / 8) + (ty >> 1)) * tiles_x + tx) * 4) + 2 * (ty & 1);
        if (has_next) {
            if (PF) {
#pragma unroll
                for (int k = 0; k < NBU; ++k) { rb[k][0] = rbn[k][0]; rb[k][1] = rbn[k][1]; }
                rg[0] = rgn[0]; rg[1] = rgn[1];
            } else {
                fetch(tile + gridDim.x, rb, rg);
            }
        }
    }
    __syncthreads();
    if (tmax && prev_tm >= 0 && tid < 2)
        tmax[prev_tm + tid] = fmaxf(fmaxf(s_omax[0], s_omax[1]), fmaxf(s_omax[2], s_omax[3]));

    // rows of the accumulator are j = tap*4 + co: registers 4q .. 4q+3 of a lane are the 4 output channels of one tap
    // (accum / alpha: the block's partial tile accumulates over the decoder passes of a training step, see conv3x3_wgrad_f16x3_ws_kernel)
    const float inv = alpha / acc_prod;
    float* pw = part + (size_t)(blockIdx.x * KS + ks) * 9 * C * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tp = mj * 8 + 2 * q + kh;
        if (tp < 9) {
            float4* dst4 = reinterpret_cast<float4*>(pw + ((size_t)tp * C + ci) * 4);
            float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (accum) o4 = *dst4;
            *dst4 = make_float4(o4.x + acc[4 * q] * inv, o4.y + acc[4 * q + 1] * inv, o4.z + acc[4 * q + 2] * inv, o4.w + acc[4 * q + 3] * inv);
        }
    }
    __syncthreads();
    float4* s_red = reinterpret_cast<float4*>(s_b);
    s_red[tid] = bsum;
    __syncthreads();
    if (tid == 0) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < 256; ++q) { const float4 v = s_red[q]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
        float4* pb = reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * 4);
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (accum) o4 = *pb;
        *pb = make_float4(o4.x + alpha * t4.x, o4.y + alpha * t4.y, o4.z + alpha * t4.z, o4.w + alpha * t4.w);
    }
}

// part: nparts x [9][c][4], part_b: nbias_parts x [4]; out = d(pre-activation) of the last hidden layer, tmax its side buffer
hipError_t launch_dec_out_bwd_fused_f16x3(hipStream_t st, const float* a, const float* g, const void* wpk, const float* wmeta,
                                          float* out, float* tmax, float* part, float* part_b, int N, int S, int c,
                                          int* nparts, int* nbias_parts, float alpha, int accum)
{
    if (S % 16 != 0 || (c != 64 && c != 32)) return hipErrorInvalidValue;
    const int tiles_x = S / 16, tiles_y = S / 4, ntiles = N * tiles_x * tiles_y;
    static const int pf = getenv("IODINE_OUTBWD_PF") ? atoi(getenv("IODINE_OUTBWD_PF")) : 1;
    static const int capenv = getenv("IODINE_OUTBWD_CAP") ? atoi(getenv("IODINE_OUTBWD_CAP")) : 0;
    const int cap = capenv ? capenv : ((c == 64 && pf) ? 768 : 1024);   // resident blocks: 3 resp. 4 per CU, persistent
    const int blocks = ntiles < cap ? ntiles : cap;
#define IOD_LAUNCH_OUTBWD(CC, PFV)                                                                                            \
    hipLaunchKernelGGL((dec_out_bwd_fused_f16x3_kernel<CC, PFV>), dim3(blocks), dim3(256), 0, st, a, g, (const uint4*)wpk, wmeta, \
                       out, tmax, part, part_b, S, ntiles, tiles_x, tiles_y, alpha, accum)
    if (c == 64) { if (pf) IOD_LAUNCH_OUTBWD(64, true); else IOD_LAUNCH_OUTBWD(64, false); }
    else { if (pf) IOD_LAUNCH_OUTBWD(32, true); else IOD_LAUNCH_OUTBWD(32, false); }
#undef IOD_LAUNCH_OUTBWD
    *nparts = blocks * (c == 64 ? 1 : 2);
    *nbias_parts = blocks;
    return hipGetLastError();
}

// =========================================================================================
// Warp-specialised form of conv3x3_wgrad_f16x3_kernel (same arithmetic, same partial-tile output).
// The one-role kernel spends 42 % of a tile waiting for its global loads and 31 % splitting / transposing them into
// LDS; only 21 % is MFMA (tools/tile_phase_prof.md), and its 144 accumulator registers leave no room to prefetch.
// Here 512 threads = one block per CU:
//   waves 0-3  CONSUMERS  hold the 9 x 32x32 accumulators, read fragments from plane buffer (t & 1), 108 MFMAs per tile
//   waves 4-7  PRODUCERS  tile t+3 in flight (three register sets), tile t+1 split + transposed into buffer ((t+1) & 1),
//                         max |a|, max |d| of tile t+2 published for the scale choice of the next step
// One s_barrier per tile.  The tile loop is unrolled by three so that the register sets are named, not indexed.
// =========================================================================================
template <class F, int... Is>
IOD_DEVINL void iod_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
IOD_DEVINL void iod_static_for(F&& f) { iod_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The producers write the split tile in NATURAL order (per pixel: 64 hi channels | 64 lo channels, 8-byte
// stores, no rotation) and the consumers fetch K-major fragments with the gfx950 transposing LDS read
// ds_read_b64_tr_b16 (tools/experiments/tr_b16_probe.hip: in a 16-lane group lane i supplies row i/4, columns 4(i%4)..+3
// of a 4x16 halfword matrix, lane c receives column c): lane = channel, 4 consecutive pixels per read, the +-1 column
// taps are plain address offsets (no v_alignbit, no neighbour reads).  Pixel slots are padded to a stride of 64 mod 256
// bytes so that the 4 pixels x 64 bytes of one read fall into different bank quarters.
// (TR is a leftover template flag of the retired transposing-stager variant, always true.)
template <int CI, int NCO, bool TR>
__global__ __launch_bounds__(512, 2)
void conv3x3_wgrad_f16x3_ws_kernel(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ part,
                                   float* __restrict__ part_b, int S, int ntiles, int tiles_x, int tiles_y, float alpha, int accum)
{
    constexpr int MT = CI / 32, NTT = NCO / 32, KS = 4 / (MT * NTT);
    constexpr int TH = 4, HH = TH + 2;
    static_assert(TR, "only the natural-order staging + ds_read_b64_tr_b16 form is built");
    constexpr int A4 = CI / 4, D4 = NCO / 4;
    constexpr int NA_UNITS = HH * 10 * A4, ND_UNITS = TH * 8 * D4;
    constexpr int NAU = (NA_UNITS + 255) / 256, NDU = (ND_UNITS + 255) / 256;
    constexpr int RW = TH / KS;
    constexpr int AST = 2 * CI * 2 + 64, DST = 2 * NCO * 2 + 64;            // TR: bytes per pixel slot (hi | lo | pad)
    constexpr int A_BYTES = HH * 20 * AST, D_BYTES = TH * 16 * DST;
    static_assert(AST % 256 == 64 || AST % 256 == 192, "TR pixel stride must be an odd multiple of 64 bytes");
    constexpr int BUF_DW = (A_BYTES + D_BYTES) / 4;

    extern __shared__ __attribute__((aligned(16))) unsigned smem_ws[];
    float* s_max = reinterpret_cast<float*>(smem_ws + 2 * BUF_DW);          // [3][8]
    float* s_scale = s_max + 24;                                            // [2]: sa * sd of the tile in buffer b

    const int tid = threadIdx.x;
    const int lane = tid & 63, kh = lane >> 5, li = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Round 6: XCD-aware tile order (block b runs on XCD b % 8 - observed, speed only): every XCD walks one contiguous eighth of the tile
    // list and its blocks take neighbouring tiles at about the same time, so that the halo rows / columns two tiles share are L2 hits.  With
    // the plain order (tile = block + q * blocks) neighbouring tiles sat on different XCDs and every halo was fetched from memory again:
    // PMC at the cfg2 shapes 286 MB per launch against 210 MB of operands + partial tiles (profiles/r06_pmc_dsprites.json).  Needs a grid
    // that is a multiple of 8; the tile -> block map only changes WHICH partial tile a product lands in (fixed: results stay deterministic).
    const bool xcd_order = (gridDim.x & 7) == 0;
    const int bpx = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;              // tile stride of a block
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_begin = xcd_order ? (int)(blockIdx.x & 7) * per_xcd : 0;
    const int t_end = xcd_order ? min(ntiles, t_begin + per_xcd) : ntiles;
    const int t_first = t_begin + (xcd_order ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    const int my_tiles = t_first < t_end ? (t_end - 1 - t_first) / bpx + 1 : 0;
    const int nq = my_tiles;

    if (wv >= 4) {
        // ------------------------------------------------------------------ PRODUCERS ----
        const int ptid = tid - 256, pw = wv - 4;
        float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
        float sa = 1.f, sd = 1.f;
        struct Set { f32x4 ra[NAU][2]; f32x4 rd[NDU][2]; };     // native vectors: they are inline-asm operands
#define VM_WAIT(n) do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
        // the wait names every register of the set as read-write, so no use of the loaded values can be placed above it
        auto wait_set = [&](auto nc, Set& r) {
            constexpr int n = decltype(nc)::value;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory");
#pragma unroll
            for (int k = 0; k < NAU; ++k) { asm volatile("" : "+v"(r.ra[k][0])); asm volatile("" : "+v"(r.ra[k][1])); }
#pragma unroll
            for (int k = 0; k < NDU; ++k) { asm volatile("" : "+v"(r.rd[k][0])); asm volatile("" : "+v"(r.rd[k][1])); }
            __builtin_amdgcn_sched_barrier(0);
        };
        using std::integral_constant;
        // Tile loads are raw BUFFER loads: one descriptor per (tensor, slot-image) with num_records = one image, so rows
        // above / below the image are out of range and return 0 in hardware; only the two halo columns left / right of
        // the image need an explicit invalid offset.  Per load: one v_add (tile offset + the thread's constant unit
        // offset) and, for the edge columns, one v_cndmask - instead of ~15 VALU of 64-bit address arithmetic each.
        typedef int i32x4_ __attribute__((ext_vector_type(4)));
        auto make_rsrc = [&](const float* base, unsigned bytes) {
            const unsigned long long p = (unsigned long long)base;
            i32x4_ r;
            r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
            r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));          // stride 0, no swizzle
            r.z = __builtin_amdgcn_readfirstlane((int)bytes);
            r.w = 0x00020000;                                                         // raw dword buffer (gfx9 / CDNA)
            return r;
        };
#define BLOAD4(dst, voff, rsrc, imm) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(dst) : "v"(voff), "s"(rsrc), "n"(imm) : "memory")
        unsigned aoff[NAU], doff[NDU];                     // the thread's constant byte offsets inside a tile (pixel j = 0)
        bool a_left[NAU], a_right[NAU];
#pragma unroll
        for (int k = 0; k < NAU; ++k) {
            const int u = ptid + k * 256;
            const int c4 = u % A4, tt = u / A4, p = tt % 10, row = tt / 10;
            aoff[k] = u < NA_UNITS ? (unsigned)(((row * S + 2 * p) * CI + c4 * 4) * 4) : 0x80000000u;
            a_left[k] = p == 0; a_right[k] = p == 9;
        }
#pragma unroll
        for (int k = 0; k < NDU; ++k) {
            const int u = ptid + k * 256;
            const int c4 = u % D4, tt = u / D4, p = tt % 8, row = tt / 8;
            doff[k] = u < ND_UNITS ? (unsigned)(((row * S + 2 * p) * NCO + c4 * 4) * 4) : 0x80000000u;
        }
        // tile coordinates of the load stream advance by gridDim.x per step: carried additions instead of two integer
        // divisions per tile (those were ~1 k ticks of a 6 k-tick step)
        int ltx, lty, ln, lq = 0;
        {
            int t = min(t_first, ntiles - 1);                   // (a block without tiles still issues its first loads: keep them inside the tensor)
            ltx = t % tiles_x; t /= tiles_x;
            lty = t % tiles_y; ln = t / tiles_y;
        }
        int gsx, gsy, gsn;
        {
            int t = bpx;
            gsx = t % tiles_x; t /= tiles_x;
            gsy = t % tiles_y; gsn = t / tiles_y;
        }
        auto G = [&](int, Set& r) {                                         // called for steps 0, 1, 2, ... in order
            const int tx = ltx, ty = lty, n = ln;
            if (lq + 1 < nq) {                                              // steps past the end re-read the last tile
                ltx += gsx;
                const int cx = ltx >= tiles_x ? 1 : 0;
                ltx -= cx ? tiles_x : 0;
                lty += gsy + cx;
                const int cy = lty >= tiles_y ? 1 : 0;
                lty -= cy ? tiles_y : 0;
                ln += gsn + cy;
            }
            ++lq;
            const i32x4_ ra_rsrc = make_rsrc(a + (size_t)n * S * S * CI, (unsigned)(S * S * CI * 4));
            const i32x4_ rd_rsrc = make_rsrc(d + (size_t)n * S * S * NCO, (unsigned)(S * S * NCO * 4));
            // first halo pixel of the tile is (ty*TH - 1, tx*16 - 2); negative offsets wrap above num_records -> 0
            const unsigned a_tile = (unsigned)(((ty * TH - 1) * S + tx * 16 - 2) * CI * 4);
            const unsigned d_tile = (unsigned)(((ty * TH) * S + tx * 16) * NCO * 4);
            const bool at_left = tx == 0, at_right = tx == tiles_x - 1;     // block-uniform
            // SGPRs written by the SALU need 5 wait states before a VMEM instruction reads them; hipcc's hazard recognizer
            // does not look inside inline asm
            asm volatile("s_nop 4" :: "s"(ra_rsrc), "s"(rd_rsrc) : "memory");
#pragma unroll
            for (int k = 0; k < NAU; ++k) {
                const bool bad = (a_left[k] && at_left) || (a_right[k] && at_right);
                const unsigned vo = bad ? 0x80000000u : aoff[k] + a_tile;   // both pixels of an edge unit are outside
                BLOAD4(r.ra[k][0], vo, ra_rsrc, 0);
                BLOAD4(r.ra[k][1], vo, ra_rsrc, CI * 4);
            }
#pragma unroll
            for (int k = 0; k < NDU; ++k) {
                const unsigned vo = doff[k] + d_tile;
                BLOAD4(r.rd[k][0], vo, rd_rsrc, 0);
                BLOAD4(r.rd[k][1], vo, rd_rsrc, NCO * 4);
            }
        };
        // The loads are inline asm, so hipcc does not count them: its own vmcnt bookkeeping would drain vmcnt(0) at the
        // loop header (every third tile) and expose the latency the three register sets are there to hide.  Loads return
        // in order; a set is (2 NAU + 2 NDU) loads, so "at most n younger sets outstanding" = vmcnt(n * SET_LOADS).
        constexpr int SET_LOADS = 2 * NAU + 2 * NDU;
        auto MAXPUB = [&](int q, int slot, const Set& r) {
            float ma = 0.f, md = 0.f;
#pragma unroll
            for (int k = 0; k < NAU; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    ma = fmaxf(ma, fmaxf(fmaxf(fabsf(r.ra[k][j].x), fabsf(r.ra[k][j].y)), fmaxf(fabsf(r.ra[k][j].z), fabsf(r.ra[k][j].w))));
            const bool real = q < nq;                                       // block-uniform: the bias sums count each tile once
#pragma unroll
            for (int k = 0; k < NDU; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    md = fmaxf(md, fmaxf(fmaxf(fabsf(r.rd[k][j].x), fabsf(r.rd[k][j].y)), fmaxf(fabsf(r.rd[k][j].z), fabsf(r.rd[k][j].w))));
                    if (real) { bsum.x += r.rd[k][j].x; bsum.y += r.rd[k][j].y; bsum.z += r.rd[k][j].z; bsum.w += r.rd[k][j].w; }
                }
            ma = wave_max_f32(ma);
            md = wave_max_f32(md);
            if (lane == 0) { s_max[slot * 8 + pw] = ma; s_max[slot * 8 + 4 + pw] = md; }
        };
        auto WIN = [&](int q, int slot, const Set& r) {
            const float* m = s_max + slot * 8;
            const float ma = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])), md = fmaxf(fmaxf(m[4], m[5]), fmaxf(m[6], m[7]));
            sa = tile_scale(ma, sa);
            sd = tile_scale(md, sd);
            if (ptid == 0) s_scale[q & 1] = sa * sd;
            if constexpr (TR) {
                unsigned char* sb = reinterpret_cast<unsigned char*>(smem_ws + (q & 1) * BUF_DW);
                auto split_store = [&](const f32x4 w, float scale, unsigned char* dst, int lo_off) {
                    const float x = w.x * scale, y = w.y * scale, z = w.z * scale, t = w.w * scale;
                    typedef __fp16 h2_ __attribute__((ext_vector_type(2)));
                    const h2_ h01 = __builtin_amdgcn_cvt_pkrtz(x, y), h23 = __builtin_amdgcn_cvt_pkrtz(z, t);      // (rounding toward zero = the 11 leading bits)
                    const h2_ l01 = __builtin_amdgcn_cvt_pkrtz(x - (float)h01.x, y - (float)h01.y), l23 = __builtin_amdgcn_cvt_pkrtz(z - (float)h23.x, t - (float)h23.y);   // v_fma_mix_f32
                    uint2 hi, lo;
                    __builtin_memcpy(&hi.x, &h01, 4); __builtin_memcpy(&hi.y, &h23, 4);
                    __builtin_memcpy(&lo.x, &l01, 4); __builtin_memcpy(&lo.y, &l23, 4);
                    *reinterpret_cast<uint2*>(dst) = hi;
                    *reinterpret_cast<uint2*>(dst + lo_off) = lo;
                };
#pragma unroll
                for (int k = 0; k < NAU; ++k) {
                    const int u = ptid + k * 256;
                    if (u < NA_UNITS) {
                        const int c4 = u % A4, tt = u / A4, p = tt % 10, row = tt / 10;
                        unsigned char* dst = sb + (row * 20 + 2 * p) * AST + c4 * 8;
                        split_store(r.ra[k][0], sa, dst, CI * 2);
                        split_store(r.ra[k][1], sa, dst + AST, CI * 2);
                    }
                }
#pragma unroll
                for (int k = 0; k < NDU; ++k) {
                    const int u = ptid + k * 256;
                    if (u < ND_UNITS) {
                        const int c4 = u % D4, tt = u / D4, p = tt % 8, row = tt / 8;
                        unsigned char* dst = sb + A_BYTES + (row * 16 + 2 * p) * DST + c4 * 8;
                        split_store(r.rd[k][0], sd, dst, NCO * 2);
                        split_store(r.rd[k][1], sd, dst + DST, NCO * 2);
                    }
                }
                return;
            }
        };
        Set R0, R1, R2;
        TP_DECL;
        G(0, R0); G(1, R1); G(2, R2);
        wait_set(integral_constant<int, 2 * SET_LOADS>{}, R0);
        MAXPUB(0, 0, R0);
        __builtin_amdgcn_s_waitcnt(0xc07f);                                 // lgkmcnt(0): own LDS writes retired
        __builtin_amdgcn_s_barrier();                                       // P1: max(0) visible
        WIN(0, 0, R0);
        wait_set(integral_constant<int, SET_LOADS>{}, R1);
        MAXPUB(1, 1, R1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                                       // P2: tile 0 staged, max(1) visible
        // iteration q: request tile q+3, stage tile q+1, publish max(q+2)
        auto iteration = [&](int q, int sl, Set& Rnext, Set& Rmax, Set& Rload) {
            TP_STAMP(7);
            G(q + 3, Rload);
            TP_STAMP(2);                                                    // [2] producer: issue of the tile loads
            wait_set(integral_constant<int, 2 * SET_LOADS>{}, Rnext);        // (tile q+1 arrived during the previous step)
            WIN(q + 1, (sl + 1) % 3, Rnext);
            TP_STAMP(3);                                                    // [3] producer: split + transposed LDS writes
            wait_set(integral_constant<int, SET_LOADS>{}, Rmax);             // tile q+2: requested two steps ago
            MAXPUB(q + 2, (sl + 2) % 3, Rmax);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            TP_STAMP(4);                                                    // [4] producer: wait for tile q+2, max
            __builtin_amdgcn_s_barrier();
            TP_STAMP(5);                                                    // [5] producer: barrier wait
        };
        for (int q = 0; q < nq; q += 3) {
            iteration(q, 0, R1, R2, R0);
            if (q + 1 < nq) iteration(q + 1, 1, R2, R0, R1);
            if (q + 2 < nq) iteration(q + 2, 2, R0, R1, R2);
        }
#ifdef IODINE_TILE_PROF
        if (ptid == 0 && blockIdx.x < TP_MAXBLK) for (int i_ = 2; i_ < 8; ++i_) g_wgrad_prof[blockIdx.x * 8 + i_] = tp_acc[i_];
#endif
        VM_WAIT(0);
#undef VM_WAIT
#undef BLOAD4
        __builtin_amdgcn_s_barrier();                                       // F1: consumers are done with the planes
        reinterpret_cast<float4*>(smem_ws)[ptid] = bsum;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                                       // F2
        return;
    }

    // ---------------------------------------------------------------------- CONSUMERS ----
    const int mi = wv % MT, ni = (wv / MT) % NTT, ks = wv / (MT * NTT);
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float acc_prod = 1.f;
    __builtin_amdgcn_s_barrier();                                           // P1
    __builtin_amdgcn_s_barrier();                                           // P2
    TP_DECL;
    for (int q = 0; q < nq; ++q) {
        const float prod = s_scale[q & 1];
        if (prod != acc_prod) {                                             // block-uniform; exact (powers of two)
            const float r = prod / acc_prod;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[tp][e] *= r;
            acc_prod = prod;
        }
        if constexpr (TR) {
            // lane -> (16-lane group g: K half kh = g >> 1, channel half g & 1; i = lane & 15: pixel i >> 2, channel quad i & 3)
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
            const int g = lane >> 4, i16 = lane & 15;
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)smem_ws + (q & 1) * BUF_DW * 4;
            const int row0 = RW == 1 ? ks : 0;                          // RW == 1: this wave's single row through the base address
            const unsigned a_addr = lds0 + (row0 * 20 + 8 * (g >> 1) + (i16 >> 2)) * AST + (mi * 32 + 16 * (g & 1) + 4 * (i16 & 3)) * 2;
            const unsigned d_addr = lds0 + A_BYTES + (row0 * 16 + 8 * (g >> 1) + (i16 >> 2)) * DST + (ni * 32 + 16 * (g & 1) + 4 * (i16 & 3)) * 2;
            struct FragA { u32x2_ v[2][3][2]; };                         // [term][dx][pixel half]
            struct FragB { u32x2_ v[2][2]; };                            // [term][pixel half]
#define IOD_TRR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
            auto loadA = [&, a_addr](auto rc, auto dyc, FragA& f) {
                constexpr int rrow = decltype(rc)::value + decltype(dyc)::value;
                const unsigned aa = a_addr;
#pragma unroll
                for (int term = 0; term < 2; ++term)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            // (constant-folded after unrolling: the offset must be an immediate)
                            switch (term * 6 + dx * 2 + h) {
#define IOD_CASE(T_, DX_, H_) case T_ * 6 + DX_ * 2 + H_: IOD_TRR(f.v[T_][DX_][H_], aa, (rrow * 20 + DX_ + 1 + 4 * H_) * AST + T_ * CI * 2); break;
                                IOD_CASE(0, 0, 0) IOD_CASE(0, 0, 1) IOD_CASE(0, 1, 0) IOD_CASE(0, 1, 1) IOD_CASE(0, 2, 0) IOD_CASE(0, 2, 1)
                                IOD_CASE(1, 0, 0) IOD_CASE(1, 0, 1) IOD_CASE(1, 1, 0) IOD_CASE(1, 1, 1) IOD_CASE(1, 2, 0) IOD_CASE(1, 2, 1)
#undef IOD_CASE
                            }
                        }
            };
            auto loadB = [&, d_addr](auto rc, FragB& f) {
                constexpr int rrow = decltype(rc)::value;
                const unsigned da = d_addr;
                IOD_TRR(f.v[0][0], da, (rrow * 16) * DST);
                IOD_TRR(f.v[0][1], da, (rrow * 16 + 4) * DST);
                IOD_TRR(f.v[1][0], da, (rrow * 16) * DST + NCO * 2);
                IOD_TRR(f.v[1][1], da, (rrow * 16 + 4) * DST + NCO * 2);
            };
#undef IOD_TRR
            auto mma = [&](auto dyc, FragA& fa, FragB& fb) {
                constexpr int dy = decltype(dyc)::value;
                h16x8 B[2], A[2][3];
#pragma unroll
                for (int term = 0; term < 2; ++term) {
                    const uint4 vb = make_uint4(fb.v[term][0].x, fb.v[term][0].y, fb.v[term][1].x, fb.v[term][1].y);
                    __builtin_memcpy(&B[term], &vb, 16);
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const uint4 va = make_uint4(fa.v[term][dx][0].x, fa.v[term][dx][0].y, fa.v[term][dx][1].x, fa.v[term][dx][1].y);
                        __builtin_memcpy(&A[term][dx], &va, 16);
                    }
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tap = dy * 3 + dx;
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][dx], B[0], acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][dx], B[1], acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][dx], B[0], acc[tap], 0, 0, 0);
                }
            };
            using std::integral_constant;
            // fragments are double-buffered: the reads of step s+1 are issued before the MFMAs of step s; LDS returns in
            // order, so waiting until only the newer step's reads are outstanding retires the older step's
            FragA fa0, fa1;
            FragB fb0, fb1;
            auto wait_frags = [&](auto nc, FragA& fa, FragB& fb) {
                constexpr int nleft = decltype(nc)::value;
                asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(nleft) : "memory");
#pragma unroll
                for (int term = 0; term < 2; ++term) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) { asm volatile("" : "+v"(fa.v[term][dx][0])); asm volatile("" : "+v"(fa.v[term][dx][1])); }
                    asm volatile("" : "+v"(fb.v[term][0])); asm volatile("" : "+v"(fb.v[term][1]));
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // rows of this wave: RW == 4: rows 0..3 (compile-time offsets); RW == 1: row ks through the base addresses
            static_assert(RW == 4 || RW == 1, "rows per consumer wave");
            // step list (row, dy): even steps use (fa0), odd steps (fa1); B fragments alternate per row
            auto run = [&]() {
                constexpr int NROW = RW;
                loadB(integral_constant<int, 0>{}, fb0);
                loadA(integral_constant<int, 0>{}, integral_constant<int, 0>{}, fa0);
                iod_static_for<NROW>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    FragB& fbc = (r & 1) ? fb1 : fb0;
                    FragB& fbn = (r & 1) ? fb0 : fb1;
                    // dy = 0 (fa of parity (3r) & 1)
                    auto& f_0 = ((3 * r) & 1) ? fa1 : fa0;
                    auto& f_1 = ((3 * r + 1) & 1) ? fa1 : fa0;
                    auto& f_2 = ((3 * r + 2) & 1) ? fa1 : fa0;
                    loadA(rc, integral_constant<int, 1>{}, f_1);
                    wait_frags(integral_constant<int, 12>{}, f_0, fbc);
                    mma(integral_constant<int, 0>{}, f_0, fbc);
                    loadA(rc, integral_constant<int, 2>{}, f_2);
                    wait_frags(integral_constant<int, 12>{}, f_1, fbc);
                    mma(integral_constant<int, 1>{}, f_1, fbc);
                    if constexpr (r + 1 < NROW) {
                        loadB(integral_constant<int, (r + 1 < NROW ? r + 1 : 0)>{}, fbn);
                        loadA(integral_constant<int, (r + 1 < NROW ? r + 1 : 0)>{}, integral_constant<int, 0>{}, f_1);   // step 3r+3 has the parity of 3r+1
                        wait_frags(integral_constant<int, 15>{}, f_2, fbc);   // (16 newer reads; lgkmcnt is a 4-bit counter)
                    } else {
                        wait_frags(integral_constant<int, 0>{}, f_2, fbc);
                    }
                    mma(integral_constant<int, 2>{}, f_2, fbc);
                });
            };
            run();
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        TP_STAMP(0);                                                        // [0] consumer: fragment reads + MFMAs of a tile
        __builtin_amdgcn_s_barrier();
        TP_STAMP(1);                                                        // [1] consumer: barrier wait
    }
#ifdef IODINE_TILE_PROF
    if (tid == 0 && blockIdx.x < TP_MAXBLK) for (int i_ = 0; i_ < 2; ++i_) g_wgrad_prof[blockIdx.x * 8 + i_] = tp_acc[i_];
#endif
    constexpr int NCOP = NTT * 32;
    // Round 5: the partial tile of THIS block can be kept across the T + 1 decoder passes of a training step (accum: add alpha x this
    // launch to what the block left there in the previous pass; alpha = the pass's loss weight -w_i / B), so that the fixed-order
    // reduction over the blocks runs once per layer and step instead of once per launch (18 -> 3 reduce launches per cfg3 step).
    const float inv = alpha / acc_prod;
    if constexpr (KS > 1) {
        // Round 6 (C = 32: the four consumer waves each own one ROW of the tile and a full set of the nine 32 x 32 tap accumulators): the four
        // K-split accumulator sets are summed through LDS - fixed order wave 0 .. 3 - before ONE partial tile per block is written, instead of
        // four (37.7 -> 9.4 MB of partial tiles per cfg2 launch, and the fixed-order reduction behind it reads a quarter).  Three rounds of three
        // taps (4 waves x 3 x 16 x 64 floats = 48 KB per round, behind the 4 KB the producers' bias sums use); the producer waves have left after F2,
        // terminated waves do not take part in s_barrier.
        __builtin_amdgcn_s_barrier();                                       // F1
        __builtin_amdgcn_s_barrier();                                       // F2: producers' bias sums are in LDS
        if (tid < D4) {
            const float4* s_red = reinterpret_cast<const float4*>(smem_ws);
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = tid; j < 256; j += D4) { const float4 v = s_red[j]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
            float4* pb = reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * NCO + tid * 4);
            float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (accum) o4 = *pb;
            *pb = make_float4(o4.x + alpha * t4.x, o4.y + alpha * t4.y, o4.z + alpha * t4.z, o4.w + alpha * t4.w);
        }
        static_assert(KS == 4 && MT == 1 && NTT == 1, "K-split reduction is written for the 32 x 32 instance");
        float* s_sum = reinterpret_cast<float*>(smem_ws) + 1024;            // [wave][3 taps x 16 regs][64 lanes]
        float* pwb = part + ((size_t)blockIdx.x * 9) * CI * NCOP;
#pragma unroll
        for (int rd = 0; rd < 3; ++rd) {
#pragma unroll
            for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_sum[(ks * 48 + tl * 16 + r) * 64 + lane] = nq > 0 ? acc[rd * 3 + tl][r] * inv : 0.f;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int ent = ks * 12 + e, tl = ent >> 4, r = ent & 15;
                const float v = ((s_sum[(0 * 48 + ent) * 64 + lane] + s_sum[(1 * 48 + ent) * 64 + lane]) + s_sum[(2 * 48 + ent) * 64 + lane]) + s_sum[(3 * 48 + ent) * 64 + lane];
                const int cr = (r & 3) + 8 * (r >> 2) + 4 * kh;
                float* dst = pwb + ((size_t)(rd * 3 + tl) * CI + cr) * NCOP + li;
                *dst = accum ? *dst + v : v;
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    float* pw = part + ((size_t)(blockIdx.x * KS + ks) * 9) * CI * NCOP;
    if (accum) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = pw[((size_t)tap * CI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * NCOP + ni * 32 + li];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                pw[((size_t)tap * CI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * NCOP + ni * 32 + li] = old[r] + (nq > 0 ? acc[tap][r] * inv : 0.f);
        }
    } else {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            pw[((size_t)tap * CI + cr) * NCOP + ni * 32 + li] = nq > 0 ? acc[tap][r] * inv : 0.f;
        }
    }
    __builtin_amdgcn_s_barrier();                                           // F1
    __builtin_amdgcn_s_barrier();                                           // F2: producers' bias sums are in LDS
    if (tid < D4) {
        const float4* s_red = reinterpret_cast<const float4*>(smem_ws);
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = tid; j < 256; j += D4) { const float4 v = s_red[j]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
        float4* pb = reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * NCO + tid * 4);
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (accum) o4 = *pb;
        *pb = make_float4(o4.x + alpha * t4.x, o4.y + alpha * t4.y, o4.z + alpha * t4.z, o4.w + alpha * t4.w);
    }
}

template <int CI, int NCO, bool TR>
static hipError_t launch_wgrad_f16_ws_inst(hipStream_t st, const float* a, const float* d, float* part, float* part_b,
                                           int N, int S, int* nparts, int* ncop, int* nbias, float alpha, int accum)
{
    constexpr int MT = CI / 32, NTT = NCO / 32, KS = 4 / (MT * NTT);
    constexpr size_t buf = true ? (size_t)6 * 20 * (2 * CI * 2 + 64) + (size_t)4 * 16 * (2 * NCO * 2 + 64)
                              : (size_t)(2 * CI * 76 + 2 * NCO * 36) * 4;
    constexpr size_t lds = 2 * buf + 26 * 4 + 32;
    static std::atomic<unsigned> attr_devs{0};                             // devices this instance is configured on
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_wgrad_f16x3_ws_kernel<CI, NCO, TR>, (int)lds, attr_devs); e != hipSuccess) return e;
    const int tiles_x = S / 16, tiles_y = S / 4, ntiles = N * tiles_x * tiles_y;
    const int blocks = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL((conv3x3_wgrad_f16x3_ws_kernel<CI, NCO, TR>), dim3(blocks), dim3(512), lds, st, a, d, part, part_b, S,
                       ntiles, tiles_x, tiles_y, alpha, accum);
#ifdef IODINE_TILE_PROF
    {
        std::vector<unsigned> hp((size_t)blocks * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_wgrad_prof), hp.size() * sizeof(unsigned));
        static const char* names[8] = {"C:mfma-phase", "C:barrier", "P:load-issue", "P:split+lds-write", "P:load-wait+max", "P:barrier", "-", "P:loop-edge"};
        double sum[8] = {0};
        for (int b2 = 0; b2 < blocks; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[(size_t)b2 * 8 + i];
        fprintf(stderr, "[wgrad ws prof <%d,%d>] memtime ticks per TILE:", CI, NCO);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f |", names[i], sum[i] / (double)ntiles);
        fprintf(stderr, "\n");
    }
#endif
    *nparts = blocks;                                                       // (round 6: the KS K-split sets of a block are summed before the write)
    *ncop = NTT * 32;
    *nbias = blocks;
    return hipGetLastError();
}

// variant 1: transposing stagers + v_alignbit shifts; variant 2: natural-order staging + ds_read_b64_tr_b16
hipError_t launch_conv3x3_wgrad_f16x3_ws(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N,
                                         int S, int ci, int nco, int* nparts, int* ncop, int* nbias_parts, float alpha, int accum)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
    if (ci == 64 && nco == 64) return launch_wgrad_f16_ws_inst<64, 64, true>(st, a, d, part, part_b, N, S, nparts, ncop, nbias_parts, alpha, accum);
    if (ci == 32 && nco == 32) return launch_wgrad_f16_ws_inst<32, 32, true>(st, a, d, part, part_b, N, S, nparts, ncop, nbias_parts, alpha, accum);
    return hipErrorInvalidValue;
}
