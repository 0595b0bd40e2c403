// Generic STRIDE-2 convolutions (any odd kernel size 3 / 5 / 7, channel counts that are multiples of 4) on the exact-fp32 matrix pipe:
// forward, data gradient and weight gradient of the refinement stack when it is off the tuned path - REF.KERNEL_SIZE 5 of the reference's
// own configs/test.yaml:40 (RefinementNetwork.mlc, lib/modeling/iodine.py:459,480: conv k s2 p k//2 + ELU, and its autograd).  Round 5;
// until then these three ran on the scalar tier of kernels_generic.hip (one thread per output element): 76 of the 118 ms of a test.yaml
// training step at batch 32.  gfx950 only.
//
// Forward and data gradient are ONE gather-GEMM kernel over v_mfma_f32_16x16x4_f32 (A = weights: 16 output channels x 4 reduction channels,
// B = 16 pixels x the same 4 channels, so a lane ends up with 4 consecutive channels of one pixel = one 16-byte store):
//   * a block owns 4 x 16 output positions, wave w one row of 16; the reduction runs over chunks of CCH channels: the input halo of the
//     chunk and the weights of the chunk's taps are staged in LDS (pixel stride CCH + 2 or + 4 floats and channel-row stride N + 16: the 16
//     pixels x 4 channels / 16 channels x 4 channels a wave reads per operand are conflict-free), one ds_read_b32 per operand and MFMA;
//   * forward (MODE 0): position (oy, ox) reads halo pixel (2 r + ky, 2 p + kx) for every tap;
//   * data gradient (MODE 1): the four parity classes of fine pixels are four stride-1 gathers over the coarse gradient with the taps of
//     matching parity (fine y = 2 Y + py receives tap ky from coarse row Y + (py + pad - ky) / 2) - no work on structural zeros; grid.y =
//     class; the result is multiplied by ELU'(aux) and stored to the fine pixel.
// The weight gradient is a K = pixels GEMM on the same instruction (A = input channels x 4 pixels, B = 4 pixels x output channels): a block
// owns one kernel ROW ky and four (16 ci, 16 co) pairs (one per wave: k accumulators), walks its slice of 4 x 16-position slabs in a fixed
// order with both operands staged per slab, and writes one partial tile per slice in the layout gen_conv_wgrad_reduce_kernel sums
// ([slice][tap][ci][co] + [co] bias partial).  Everything is deterministic; products are IEEE fp32, accumulation fp32.
#include "common.h"
#include <algorithm>

namespace {

IOD_DEVINL float gs2_elu(float v) { return v > 0.f ? v : expm1f(v); }

constexpr int GS2_TR = 4, GS2_TC = 16;                       // output positions per block: rows x columns (wave = row)

struct Gs2Geom {
    int Sin;            // spatial size of the staged tensor (forward: the fine input; data gradient: the coarse gradient)
    int Sout;           // spatial size of the position grid the tiles cover (both modes: the coarse size)
    int Sf;             // fine size (data gradient: the output's size)
    int K, ldk;         // reduction channels / channel stride of the staged tensor
    int Nn, ldn;        // output channels / channel stride of the output
    int ks;             // kernel size
    int w_tap, w_k, w_n;  // strides of W(tap, k, n) in the packed [tap][ci][co] weight
    int cch;            // reduction channels per staged chunk (4, 8 or 16)
    unsigned kmask;     // bit i: reduction channels 4 i .. 4 i + 3 can be non-zero (K <= 128; all ones otherwise) - zero groups are skipped
    int elu;
    int tiles_x, tiles_y;
};

// MODE 0: out[n][oy][ox][:] = act(bias + sum_{tap, k} in[n][2 oy + ky - pad][2 ox + kx - pad][k] W(tap, k, :))
// MODE 1: out[n][2 Y + py][2 X + px][:] = ELU'(aux) * sum_{taps of the class, k} in[n][Y + oy][X + ox][k] W(tap, k, :)
template <int MODE, int C4N, int NR>                          // C4N: 4-channel steps per staged chunk (1, 2 or 4); NR: position rows per wave (1 or
__global__ __launch_bounds__(256, 2)                          // 2: a block of 8 x 16 positions stages the chunk's weights half as often per output)
void gen_s2_conv_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                        const float* __restrict__ aux, float* __restrict__ out, Gs2Geom g)
{
    extern __shared__ __attribute__((aligned(16))) float smem_gs2[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lp = lane & 15, lq = lane >> 4;
    const int ks = g.ks, pad = ks >> 1;
    constexpr int cch = 4 * C4N;
    const int H = (pad + 1) >> 1;                              // data gradient: halo radius of the coarse gather
    constexpr int TRB = GS2_TR * NR;                           // position rows of the block: wave w owns rows w, w + 4
    const int HR = MODE == 0 ? 2 * (TRB - 1) + ks : TRB + 2 * H;
    const int HC = MODE == 0 ? 2 * (GS2_TC - 1) + ks : GS2_TC + 2 * H;
    const int PS = cch + (MODE == 0 ? 2 : 4);                  // floats per staged pixel
    const int n0 = blockIdx.z * 64;                            // first output channel of this block
    const int nn = min(64, g.Nn - n0), NG = (nn + 15) >> 4;
    const int NP = NG * 16 + 16;                               // floats per staged weight row (one reduction channel of one tap)
    const int cls = MODE == 1 ? (int)blockIdx.y : 0, py = cls >> 1, px = cls & 1;
    // taps of this block: forward all ks x ks; data gradient those with ky = (py + pad) mod 2, kx = (px + pad) mod 2
    const int ky0 = MODE == 1 ? (py + pad) & 1 : 0, kx0 = MODE == 1 ? (px + pad) & 1 : 0, kst = MODE == 1 ? 2 : 1;
    const int nky = (ks - ky0 + kst - 1) / kst, nkx = (ks - kx0 + kst - 1) / kst, ntap = nky * nkx;
    float* s_in = smem_gs2;                                    // [HR * HC][PS]
    float* s_w = smem_gs2 + ((HR * HC * PS + 3) & ~3);         // [ntap][cch][NP]

    int t = blockIdx.x;
    const int tx = t % g.tiles_x; t /= g.tiles_x;
    const int ty = t % g.tiles_y;
    const int n = t / g.tiles_y;
    const int r0 = ty * TRB, c0 = tx * GS2_TC;                // first position of the tile
    const int hy0 = MODE == 0 ? 2 * r0 - pad : r0 - H, hx0 = MODE == 0 ? 2 * c0 - pad : c0 - H;
    const float* in_n = in + (size_t)n * g.Sin * g.Sin * g.ldk;

    f32x4 acc[NR][4];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    int lane_pix[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) lane_pix[i] = MODE == 0 ? (2 * (wv + 4 * i) * HC + 2 * lp) : ((wv + 4 * i) * HC + lp);
    constexpr int c4n = C4N;
    for (int k0 = 0; k0 < g.K; k0 += cch) {
        const unsigned cm = k0 < 128 ? (g.kmask >> (k0 >> 2)) & ((1u << C4N) - 1u) : ~0u;      // live 4-channel steps of this chunk (block-uniform)
        if (!cm) continue;                                     // nothing but zero weights x zero inputs in this chunk (absent encoding channels)
        __syncthreads();                                       // the previous chunk has been read
        // ---- halo chunk: float4 (pixel, channel quad) ----
        for (int i = tid; i < HR * HC * c4n; i += 256) {
            const int c4 = i % c4n, pix = i / c4n, hr = pix / HC, hc = pix - hr * HC;
            const int gy = hy0 + hr, gx = hx0 + hc, ch = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gy < (unsigned)g.Sin && (unsigned)gx < (unsigned)g.Sin && ch < g.K) {
                v = *reinterpret_cast<const float4*>(in_n + ((size_t)gy * g.Sin + gx) * g.ldk + ch);
                if (ch + 3 >= g.K) {                           // ragged channel count (17 of 20): what lies past K is not part of the conv
                    if (ch + 1 >= g.K) v.y = 0.f;
                    if (ch + 2 >= g.K) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            float* d = s_in + pix * PS + 4 * c4;               // (PS is even: 8-byte aligned)
            *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
        }
        // ---- weights of the chunk: s_w[j][kk][nl] = W(tap_j, k0 + kk, n0 + nl), zero outside ----
        // (16-byte loads along the contiguous axis of the packed weight: n for the forward conv, k for the data gradient)
        if (MODE == 0) {
            const int n4 = NG * 4;
            for (int i = tid; i < ntap * cch * n4; i += 256) {
                const int nl = 4 * (i % n4), r = i / n4, kk = r % cch, j = r / cch;
                const int ky = ky0 + kst * (j / nkx), kx = kx0 + kst * (j % nkx);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + kk < g.K && nl < nn)
                    v = *reinterpret_cast<const float4*>(wt + (size_t)(ky * ks + kx) * g.w_tap + (size_t)(k0 + kk) * g.w_k + n0 + nl);
                *reinterpret_cast<float4*>(s_w + (j * cch + kk) * NP + nl) = v;
            }
        } else {
            const int nr = NG * 16;
            for (int i = tid; i < ntap * nr * c4n; i += 256) {
                const int c4 = i % c4n, r = i / c4n, nl = r % nr, j = r / nr;
                const int ky = ky0 + kst * (j / nkx), kx = kx0 + kst * (j % nkx);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + 4 * c4 < g.K && nl < nn)             // (K = the conv's output channels: a multiple of 4)
                    v = *reinterpret_cast<const float4*>(wt + (size_t)(ky * ks + kx) * g.w_tap + (size_t)(n0 + nl) * g.w_n + k0 + 4 * c4);
                float* d = s_w + (j * cch + 4 * c4) * NP + nl;
                d[0] = v.x; d[NP] = v.y; d[2 * NP] = v.z; d[3 * NP] = v.w;
            }
        }
        __syncthreads();
        // ---- MFMAs: per tap and 4-channel step one B read (pixels) and NG A reads (weights) ----
        // (the operands of tap j + 1 are read while the MFMAs of tap j issue)
        float bv[2][NR][C4N], av[2][C4N][4];
        int jy = 0, jx = 0;
        auto load_tap = [&](int j, float (&b)[NR][C4N], float (&a)[C4N][4]) {
            int toff;
            if (MODE == 0) toff = jy * HC + jx;
            else {
                const int ky = ky0 + 2 * jy, kx = kx0 + 2 * jx;
                toff = (H + ((py + pad - ky) >> 1)) * HC + H + ((px + pad - kx) >> 1);      // (arithmetic shift: the numerator is even)
            }
            const float* pa = s_w + (j * cch + lq) * NP + lp;
#pragma unroll
            for (int k4 = 0; k4 < C4N; ++k4) {
                if (!((cm >> k4) & 1u)) continue;
#pragma unroll
                for (int i = 0; i < NR; ++i) b[i][k4] = s_in[(lane_pix[i] + toff) * PS + lq + 4 * k4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[k4][q] = q < NG ? pa[4 * k4 * NP + 16 * q] : 0.f;
            }
            if (++jx == nkx) { jx = 0; ++jy; }
        };
        auto mma_tap = [&](const float (&b)[NR][C4N], const float (&a)[C4N][4]) {
#pragma unroll
            for (int k4 = 0; k4 < C4N; ++k4) {
                if (!((cm >> k4) & 1u)) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < NR; ++i)
                        if (q < NG) acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k4][q], b[i][k4], acc[i][q], 0, 0, 0);
            }
        };
        load_tap(0, bv[0], av[0]);
        for (int j = 0; j < ntap; j += 2) {
            if (j + 1 < ntap) load_tap(j + 1, bv[1], av[1]);
            mma_tap(bv[0], av[0]);
            if (j + 1 < ntap) {
                if (j + 2 < ntap) load_tap(j + 2, bv[0], av[0]);
                mma_tap(bv[1], av[1]);
            }
        }
    }
    // ---- epilogue: lane = 4 consecutive channels (16 q + 4 lq ..) of positions (r0 + wv + 4 i, c0 + lp) ----
#pragma unroll
    for (int i = 0; i < NR; ++i) {
    const int oy = r0 + wv + 4 * i, ox = c0 + lp;
    if (oy >= g.Sout || ox >= g.Sout) continue;
    size_t opix;
    if (MODE == 0) opix = ((size_t)n * g.Sout + oy) * g.Sout + ox;
    else {
        const int fy = 2 * oy + py, fx = 2 * ox + px;
        if (fy >= g.Sf || fx >= g.Sf) continue;
        opix = ((size_t)n * g.Sf + fy) * g.Sf + fx;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = n0 + 16 * q + 4 * lq;
        if (q >= NG || c >= g.Nn) continue;                    // (Nn is a multiple of 4: a float4 is all in or all out)
        f32x4 v = acc[i][q];
        float* o = out + opix * g.ldn + c;
        if (MODE == 0) {
            if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + c); v += f32x4{b4.x, b4.y, b4.z, b4.w}; }
            if (g.elu) v = f32x4{gs2_elu(v.x), gs2_elu(v.y), gs2_elu(v.z), gs2_elu(v.w)};
        } else if (aux) {
            const float4 a4 = *reinterpret_cast<const float4*>(aux + opix * g.ldn + c);
            v.x *= elu1_grad_from_out(a4.x); v.y *= elu1_grad_from_out(a4.y);
            v.z *= elu1_grad_from_out(a4.z); v.w *= elu1_grad_from_out(a4.w);
        }
        *reinterpret_cast<float4*>(o) = make_float4(v.x, v.y, v.z, v.w);
    }
    }
}

inline int gs2_cch(int mode, int ks, int K, int nn, size_t* lds_out, int nr = 1)
{
    const int pad = ks / 2, H = (pad + 1) / 2, TRB = GS2_TR * nr;
    const int HR = mode == 0 ? 2 * (TRB - 1) + ks : TRB + 2 * H, HC = mode == 0 ? 2 * (GS2_TC - 1) + ks : GS2_TC + 2 * H;
    const int ntap = mode == 0 ? ks * ks : ((ks + 1) / 2) * ((ks + 1) / 2);
    const int NP = ((std::min(nn, 64) + 15) / 16) * 16 + 16;
    for (int cch = K <= 4 ? 4 : (K <= 8 ? 8 : 16); cch >= 4; cch >>= 1) {
        const size_t b = ((size_t)((HR * HC * (cch + (mode == 0 ? 2 : 4)) + 3) & ~3) + (size_t)ntap * cch * NP) * sizeof(float);
        if (b <= 72 * 1024) { *lds_out = b; return cch; }
    }
    return 0;
}

template <int MODE, int C4N, int NR>
hipError_t gs2_launch_c(hipStream_t st, dim3 grid, size_t lds, const float* in, const float* wt, const float* bias, const float* aux, float* out,
                        const Gs2Geom& g)
{
    static std::atomic<unsigned> attr_devs{0};
    if (hipError_t e = iod_set_max_lds((const void*)gen_s2_conv_kernel<MODE, C4N, NR>, 72 * 1024, attr_devs); e != hipSuccess) return e;
    hipLaunchKernelGGL((gen_s2_conv_kernel<MODE, C4N, NR>), grid, dim3(256), lds, st, in, wt, bias, aux, out, g);
    return hipGetLastError();
}
// geometry + launch: 8 x 16 positions per block (NR = 2) when the position grid has more than 4 rows and the taller halo fits beside the same
// weight chunk, else 4 x 16
template <int MODE>
hipError_t gs2_launch(hipStream_t st, int N, int nz, int ygrid, const float* in, const float* wt, const float* bias, const float* aux, float* out,
                      Gs2Geom g)
{
    size_t lds1 = 0, lds2 = 0;
    const int c1 = gs2_cch(MODE, g.ks, g.K, g.Nn, &lds1, 1), c2 = g.Sout > GS2_TR ? gs2_cch(MODE, g.ks, g.K, g.Nn, &lds2, 2) : 0;
    if (!c1) return hipErrorInvalidValue;
    const int nr = c2 == c1 ? 2 : 1;
    g.cch = c1;
    g.tiles_x = (g.Sout + GS2_TC - 1) / GS2_TC;
    g.tiles_y = (g.Sout + GS2_TR * nr - 1) / (GS2_TR * nr);
    const dim3 grid((unsigned)(N * g.tiles_x * g.tiles_y), (unsigned)ygrid, (unsigned)nz);
    const size_t lds = nr == 2 ? lds2 : lds1;
#define GS2_CASE(C4, R) if (g.cch == 4 * C4 && nr == R) return gs2_launch_c<MODE, C4, R>(st, grid, lds, in, wt, bias, aux, out, g);
    GS2_CASE(4, 2) GS2_CASE(2, 2) GS2_CASE(1, 2) GS2_CASE(4, 1) GS2_CASE(2, 1) GS2_CASE(1, 1)
#undef GS2_CASE
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// weight gradient: part[slice][tap][ci][co] (+ [co] bias partial behind the ks * ks * Ci * Co tap elements)
// grid = (ks kernel rows) x (groups of 4 (ci16, co16) pairs) x (slices)
template <int KS>
__global__ __launch_bounds__(256, 2)
void gen_s2_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ part, int N, int Si, int So,
                         int Ci, int ldc, int Co, int nsl, int npg, unsigned cimask)
{
    constexpr int ks = KS;
    extern __shared__ __attribute__((aligned(16))) float smem_gs2[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lm = lane & 15, lq = lane >> 4;
    const int pad = ks >> 1;
    int b = blockIdx.x;
    const int ky = b % ks; b /= ks;
    const int pg = b % npg;
    const int sl = b / npg;
    const int ncig = (Ci + 15) >> 4, ncog = (Co + 15) >> 4, npair = ncig * ncog;
    const int pair = pg * 4 + wv;                              // this wave's (ci group, co group); waves past the last pair idle through the MFMAs
    const bool live = pair < npair;
    const int cig = live ? pair / ncog : 0, cog = live ? pair % ncog : 0;
    const int Cip = (ncig * 16 + 31) / 32 * 32 + 8, Cop = ncog * 16 + 16;      // staged pixel strides (floats): 2 * Cip = 16 and Cop = 16 mod 64 - or 48
    const int AC = 2 * (GS2_TC - 1) + ks;                      // staged input columns of a slab row
    float* s_a = smem_gs2;                                     // [GS2_TR][AC][Cip]: the fine input rows 2 oy + ky - pad of the slab
    float* s_d = smem_gs2 + GS2_TR * AC * Cip;                 // [GS2_TR * GS2_TC][Cop]

    const int tiles_x = (So + GS2_TC - 1) / GS2_TC, tiles_y = (So + GS2_TR - 1) / GS2_TR;
    const int nslab = N * tiles_y * tiles_x;
    f32x4 acc[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};                    // bias partial: row 0 of a ones x gradient product
    const bool work = live && ((cimask >> min(cig, 31)) & 1u);       // a 16-channel group of all-zero inputs (absent encoding channels): zeros, no MFMAs
    const bool do_bias = live && ky == 0 && cig == 0;
    const int ci4 = (Ci + 3) >> 2, co4 = Co >> 2;

    for (int slab = sl; slab < nslab; slab += nsl) {
        int t = slab;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const int r0 = ty * GS2_TR, c0 = tx * GS2_TC;
        const float* in_n = in + (size_t)n * Si * Si * ldc;
        const float* d_n = dout + (size_t)n * So * So * Co;
        __syncthreads();                                       // the previous slab has been read
        for (int i = tid; i < GS2_TR * AC * ci4; i += 256) {
            const int c4 = i % ci4, pix = i / ci4, r = pix / AC, hc = pix - r * AC;
            const int gy = 2 * (r0 + r) + ky - pad, gx = 2 * c0 - pad + hc, ch = 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + r < So && (unsigned)gy < (unsigned)Si && (unsigned)gx < (unsigned)Si) {
                v = *reinterpret_cast<const float4*>(in_n + ((size_t)gy * Si + gx) * ldc + ch);
                if (ch + 3 >= Ci) {
                    if (ch + 1 >= Ci) v.y = 0.f;
                    if (ch + 2 >= Ci) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            *reinterpret_cast<float4*>(s_a + pix * Cip + ch) = v;
        }
        for (int i = tid; i < GS2_TR * GS2_TC * co4; i += 256) {
            const int c4 = i % co4, pix = i / co4, r = pix / GS2_TC, p = pix - r * GS2_TC;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + r < So && c0 + p < So) v = *reinterpret_cast<const float4*>(d_n + ((size_t)(r0 + r) * So + c0 + p) * Co + 4 * c4);
            *reinterpret_cast<float4*>(s_d + pix * Cop + 4 * c4) = v;
        }
        if (ci4 * 4 < ncig * 16)                               // channels between Ci (rounded to 4) and the last 16-group: zero rows of the product
            for (int i = tid; i < GS2_TR * AC * (ncig * 16 - ci4 * 4); i += 256) {
                const int c = ci4 * 4 + i % (ncig * 16 - ci4 * 4), pix = i / (ncig * 16 - ci4 * 4);
                s_a[pix * Cip + c] = 0.f;
            }
        __syncthreads();
        if (work || do_bias) {
            // K = positions: 4 per MFMA (lq), 16 per slab row -> 4 steps per row
#pragma unroll 1
            for (int r = 0; r < GS2_TR; ++r) {
                const float* pa = s_a + (r * AC + 2 * lq) * Cip + cig * 16 + lm;
                const float* pd = s_d + (r * GS2_TC + lq) * Cop + cog * 16 + lm;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float dv = pd[4 * s4 * Cop];
                    float av[KS];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) av[kx] = pa[(8 * s4 + kx) * Cip];
                    if (work) {
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kx], dv, acc[kx], 0, 0, 0);
                    }
                    if (do_bias) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, dv, accb, 0, 0, 0);
                }
            }
        }
    }
    if (!live) return;
    // D layout: acc[kx][i] = row (ci) 4 lq + i, column (co) lm
    const size_t per = (size_t)ks * ks * Ci * Co + Co;
    float* pp = part + (size_t)sl * per;
    const int co = cog * 16 + lm;
    if (co < Co) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ci = cig * 16 + 4 * lq + i;
                    if (ci < Ci) pp[((size_t)(ky * ks + kx) * Ci + ci) * Co + co] = acc[kx][i];
                }
        if (do_bias && lq == 0) pp[(size_t)ks * ks * Ci * Co + co] = accb[0];
    }
}

}  // namespace

// the MFMA forms apply to: stride 2, kernel size 3 / 5 / 7, channel counts / strides that are multiples of 4 (16-byte staging)
bool gen_s2_mfma_ok(int k, int K, int ldk, int Nn, int ldn)
{
    return (k == 3 || k == 5 || k == 7) && K >= 1 && (ldk & 3) == 0 && (Nn & 3) == 0 && (ldn & 3) == 0 && Nn >= 4;
}

// forward: in [N][Si][Si][ldc] (Ci of ldc channels), wt [tap][Ci][Co] -> out [N][So][So][Co], So = (Si - 1) / 2 + 1
// 4-channel groups of a per-channel mask (bit c = channel c can be non-zero), for K <= 128 channels
static unsigned gs2_kmask4(unsigned chmask, int K)
{
    if (chmask == 0xffffffffu || K > 32) return 0xffffffffu;       // (the per-channel mask has 32 bits: larger layers are never subsets)
    unsigned m = 0;
    for (int c = 0; c < K; ++c) if ((chmask >> c) & 1u) m |= 1u << (c >> 2);
    return m;
}

hipError_t launch_gen_s2_fwd(hipStream_t st, const float* in, const float* wt, const float* bias, float* out, int N, int Si, int Ci, int ldc,
                             int Co, int k, int elu, unsigned chmask)
{
    if (!gen_s2_mfma_ok(k, Ci, ldc, Co, Co)) return hipErrorInvalidValue;
    const int So = (Si - 1) / 2 + 1;
    Gs2Geom g{Si, So, Si, Ci, ldc, Co, Co, k, Ci * Co, Co, 1, 0, gs2_kmask4(chmask, Ci), elu, 0, 0};
    return gs2_launch<0>(st, N, (Co + 63) / 64, 1, in, wt, bias, nullptr, out, g);
}

// data gradient: dout [N][So][So][Co], wt [tap][ldi][Co] -> din [N][Si][Si][ldi] (Ci <= ldi channels computed) times ELU'(aux)
hipError_t launch_gen_s2_dgrad(hipStream_t st, const float* dout, const float* wt, const float* aux, float* din, int N, int Si, int Ci, int ldi,
                               int Co, int k)
{
    if (!gen_s2_mfma_ok(k, Co, Co, Ci, ldi)) return hipErrorInvalidValue;
    const int So = (Si - 1) / 2 + 1;
    // W(tap, k = co, n = ci) = wt[tap][ci][co]
    Gs2Geom g{So, So, Si, Co, Co, Ci, ldi, k, ldi * Co, 1, Co, 0, 0xffffffffu, 0, 0, 0};
    return gs2_launch<1>(st, N, (Ci + 63) / 64, 4, dout, wt, nullptr, aux, din, g);
}

// weight gradient partials: part [nsl][k * k * Ci * Co + Co]; *nsl_out slices were written (sum them with gen_conv_wgrad_reduce_kernel)
hipError_t launch_gen_s2_wgrad(hipStream_t st, const float* in, const float* dout, float* part, int N, int Si, int Ci, int ldc, int Co, int k,
                               int nsl_max, int* nsl_out, unsigned chmask)
{
    unsigned cimask = 0xffffffffu;                                  // 16-channel groups with a channel that can be non-zero
    if (chmask != 0xffffffffu && Ci <= 32) {
        cimask = 0;
        for (int c = 0; c < Ci; ++c) if ((chmask >> c) & 1u) cimask |= 1u << (c >> 4);
    }
    if (!gen_s2_mfma_ok(k, Ci, ldc, Co, Co)) return hipErrorInvalidValue;
    const int So = (Si - 1) / 2 + 1;
    const int ncig = (Ci + 15) / 16, ncog = (Co + 15) / 16, npg = (ncig * ncog + 3) / 4;
    const int Cip = (ncig * 16 + 31) / 32 * 32 + 8, Cop = ncog * 16 + 16, AC = 2 * (GS2_TC - 1) + k;
    const size_t lds = (size_t)(GS2_TR * AC * Cip + GS2_TR * GS2_TC * Cop) * sizeof(float);
    if (lds > 72 * 1024) return hipErrorInvalidValue;
    static std::atomic<unsigned> d3{0}, d5{0}, d7{0};
    if (hipError_t e = k == 3 ? iod_set_max_lds((const void*)gen_s2_wgrad_kernel<3>, 72 * 1024, d3)
                     : k == 5 ? iod_set_max_lds((const void*)gen_s2_wgrad_kernel<5>, 72 * 1024, d5)
                              : iod_set_max_lds((const void*)gen_s2_wgrad_kernel<7>, 72 * 1024, d7); e != hipSuccess) return e;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    const int nslab = N * ((So + GS2_TC - 1) / GS2_TC) * ((So + GS2_TR - 1) / GS2_TR);
    const int nsl = std::max(1, std::min(std::min(nsl_max, nslab), std::max(1, 2 * n_cu / (k * npg))));
    const dim3 grid((unsigned)(k * npg * nsl));
    if (k == 3) hipLaunchKernelGGL((gen_s2_wgrad_kernel<3>), grid, dim3(256), lds, st, in, dout, part, N, Si, So, Ci, ldc, Co, nsl, npg, cimask);
    else if (k == 5) hipLaunchKernelGGL((gen_s2_wgrad_kernel<5>), grid, dim3(256), lds, st, in, dout, part, N, Si, So, Ci, ldc, Co, nsl, npg, cimask);
    else hipLaunchKernelGGL((gen_s2_wgrad_kernel<7>), grid, dim3(256), lds, st, in, dout, part, N, Si, So, Ci, ldc, Co, nsl, npg, cimask);
    *nsl_out = nsl;
    return hipGetLastError();
}

bool gen_s2_wgrad_ok(int k, int Ci, int ldc, int Co)
{
    if (!gen_s2_mfma_ok(k, Ci, ldc, Co, Co)) return false;
    const int ncig = (Ci + 15) / 16, ncog = (Co + 15) / 16;
    const int Cip = (ncig * 16 + 31) / 32 * 32 + 8, Cop = ncog * 16 + 16, AC = 2 * (GS2_TC - 1) + k;
    return (size_t)(GS2_TR * AC * Cip + GS2_TR * GS2_TC * Cop) * sizeof(float) <= 72 * 1024;
}
bool gen_s2_fwd_ok(int k, int Ci, int ldc, int Co) { size_t l; return gen_s2_mfma_ok(k, Ci, ldc, Co, Co) && gs2_cch(0, k, Ci, Co, &l) != 0; }
bool gen_s2_dgrad_ok(int k, int Ci, int ldi, int Co) { size_t l; return gen_s2_mfma_ok(k, Co, Co, Ci, ldi) && gs2_cch(1, k, Co, Ci, &l) != 0; }
