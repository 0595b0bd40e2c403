// Generic (any odd kernel size, any channel count) fp32 convolution kernels: the FALLBACK path of the library for ARCH
// combinations the tuned gfx950 kernels are not built for - KERNEL_SIZE 5 / 7 (the reference's default DEC.KERNEL_SIZE is 5,
// lib/config/defaults.py:100; configs/test.yaml:40,44) and CONV_CHAN other than 32 / 64.  Reference call sites: nn.Conv2d + F.elu of
// MultiLayerConv (iodine.py:570-594), the output conv (iodine.py:422,435), SpatialBroadcast (iodine.py:505-540) and their autograd.
//
// Two tiers.  Stride-1 convs (the decoder: ~96 % of the FLOPs) run on exact-fp32 MFMA kernels since round 4 (further down: LDS-resident
// weight slice per 16 output channels, 16 x 16 pixel tiles, v_mfma_f32_16x16x4_f32; weight gradient on v_mfma_f32_32x32x2_f32) whenever
// the weight slice fits LDS - 5 x 5 x 64 channels does, 7 x 7 up to 32 channels.  Stride 2 (the refinement stack) runs on the MFMA kernels of
// kernels_gens2.hip since round 5.  Everything else (larger slices, channel counts that are not multiples of 4) runs on the scalar kernels
// right below: one thread per output element, plain fp32 FMAs in a fixed order, written for correctness.  All of it is deterministic.  The spatial-broadcast layer (decoder layer 0) has its own kernels (kernels_genl0.hip: the broadcast
// tensor is never built; until round 5 it was materialised as [N][P][L+2] and convolved like any other layer).  Every shipped / benchmarked configuration stays on the
// tuned path (iodine_api.cpp: `generic` is false for KERNEL_SIZE 3 with 32 / 64 channels).  Measured (MI355X, CLEVR shapes with
// DEC.KERNEL_SIZE 5, batch 4): training step 4977 -> 206 ms, reconstruct 2317 -> 96 ms against the scalar tier (round 4); 48.7 / 30.6 ms
// after round 5 (what each kernel below gained and where its remaining time goes).
//
// Layouts: activations NHWC with a channel stride `ldc` >= Ci (the 17-of-20 refinement input); weights re-packed at set_params to
// [tap = ky * k + kx][ci][co] (co fastest: coalesced over the threads of a pixel); stride s in {1, 2}, padding k / 2.
#include "common.h"

namespace {

IOD_DEVINL float gen_elu(float v) { return v > 0.f ? v : expm1f(v); }

// w [Co][Ci][k][k] -> wt [k*k][Ci][Co]
__global__ void gen_pack_weights_kernel(const float* __restrict__ w, int Co, int Ci, int kk, float* __restrict__ wt)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Co * Ci * kk) return;
    const int co = idx % Co, ci = (idx / Co) % Ci, tap = idx / (Co * Ci);
    wt[idx] = w[((size_t)co * Ci + ci) * kk + tap];
}

// out[n][oy][ox][co] = act(bias[co] + sum_{tap, ci} in[n][oy*s + ky - pad][ox*s + kx - pad][ci] * wt[tap][ci][co])
__global__ void gen_conv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                                    float* __restrict__ out, int Si, int So, int Ci, int ldc, int Co, int k, int s, int elu, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % Co);
    size_t r = idx / Co;
    const int ox = (int)(r % So); r /= So;
    const int oy = (int)(r % So);
    const size_t n = r / So;
    const int pad = k / 2;
    float acc = bias ? bias[co] : 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy * s + ky - pad;
        if (iy < 0 || iy >= Si) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox * s + kx - pad;
            if (ix < 0 || ix >= Si) continue;
            const float* ip = in + ((n * Si + iy) * Si + ix) * (size_t)ldc;
            const float* wp = wt + ((size_t)(ky * k + kx) * Ci) * Co + co;
            for (int ci = 0; ci < Ci; ++ci) acc = fmaf(ip[ci], wp[(size_t)ci * Co], acc);
        }
    }
    out[idx] = elu ? gen_elu(acc) : acc;
}

// din[n][y][x][ci] = f'(aux) * sum_{tap, co} dout[n][oy][ox][co] * wt[tap][ci][co]   with oy * s + ky - pad = y (ox likewise);
// aux = the layer's INPUT activation (an ELU output): f' = aux > 0 ? 1 : aux + 1; aux = NULL: no factor.  ldi = channel stride of din
// = input channels of the PACKED weight; Ci <= ldi channels are computed (the broadcast layer needs only its L latent channels).
__global__ void gen_conv_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ wt, const float* __restrict__ aux,
                                      float* __restrict__ din, int Si, int So, int Ci, int ldi, int Co, int k, int s, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ci = (int)(idx % Ci);
    size_t r = idx / Ci;
    const int x = (int)(r % Si); r /= Si;
    const int y = (int)(r % Si);
    const size_t n = r / Si;
    const int pad = k / 2;
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int ty = y + pad - ky;
        if (ty < 0 || ty % s != 0 || ty / s >= So) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int tx = x + pad - kx;
            if (tx < 0 || tx % s != 0 || tx / s >= So) continue;
            const float* dp = dout + ((n * So + ty / s) * So + tx / s) * (size_t)Co;
            const float* wp = wt + ((size_t)(ky * k + kx) * ldi + ci) * Co;
            for (int co = 0; co < Co; ++co) acc = fmaf(dp[co], wp[co], acc);
        }
    }
    const size_t o = ((n * Si + y) * Si + x) * (size_t)ldi + ci;
    if (aux) { const float a = aux[o]; acc *= a > 0.f ? 1.f : a + 1.f; }
    din[o] = acc;
}

// partial[slice][tap][ci][co] = sum over the slice's output pixels (fixed order) of in[...][ci] * dout[...][co];
// pseudo-tap k*k, ci = 0: the bias gradient sum dout[...][co]
__global__ void gen_conv_wgrad_partial_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ part,
                                              int N, int Si, int So, int Ci, int ldc, int Co, int k, int s, int nslice)
{
    const int kk = k * k, per = kk * Ci * Co + Co;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nslice * per) return;
    const int e = (int)(idx % per), slice = (int)(idx / per);
    const size_t npx = (size_t)N * So * So, chunk = (npx + nslice - 1) / nslice;
    const size_t q0 = (size_t)slice * chunk, q1 = q0 + chunk < npx ? q0 + chunk : npx;
    float acc = 0.f;
    if (e >= kk * Ci * Co) {
        const int co = e - kk * Ci * Co;
        for (size_t q = q0; q < q1; ++q) acc += dout[q * Co + co];
    } else {
        const int co = e % Co, ci = (e / Co) % Ci, tap = e / (Co * Ci);
        const int ky = tap / k, kx = tap % k, pad = k / 2;
        for (size_t q = q0; q < q1; ++q) {
            const int ox = (int)(q % So), oy = (int)((q / So) % So);
            const size_t n = q / ((size_t)So * So);
            const int iy = oy * s + ky - pad, ix = ox * s + kx - pad;
            if (iy < 0 || iy >= Si || ix < 0 || ix >= Si) continue;
            acc = fmaf(in[((n * Si + iy) * Si + ix) * (size_t)ldc + ci], dout[q * Co + co], acc);
        }
    }
    part[idx] = acc;
}

// gw[co][ci_dst][tap] += alpha * sum_slice partial (OIHW, ci < Ci_dst kept), gb[co] += alpha * sum_slice partial bias.  A block of 256 threads
// reduces 32 consecutive elements: eight thread groups take every eighth slice, their partial sums are added in a fixed order (one thread per
// element walking 512 slices was 0.12 ms for the output conv's 13 MB of partials).
__global__ __launch_bounds__(256) void gen_conv_wgrad_reduce_kernel(const float* __restrict__ part, int nslice, int Ci, int Ci_dst, int Co, int kk,
                                                                    float alpha, float* __restrict__ gw, float* __restrict__ gb)
{
    __shared__ float red[8][32];
    const int per = kk * Ci * Co + Co;
    const int el = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float acc = 0.f;
    if (e < per)
        for (int sl = sg; sl < nslice; sl += 8) acc += part[(size_t)sl * per + e];
    red[sg][el] = acc;
    __syncthreads();
    if (sg != 0 || e >= per) return;
#pragma unroll
    for (int g2 = 1; g2 < 8; ++g2) acc += red[g2][el];
    if (e >= kk * Ci * Co) { if (gb) gb[e - kk * Ci * Co] += alpha * acc; return; }
    const int co = e % Co, ci = (e / Co) % Ci, tap = e / (Co * Ci);
    if (ci < Ci_dst) gw[((size_t)co * Ci_dst + ci) * kk + tap] += alpha * acc;
}

// rows [rows][L] identity-embedded in [rows][L] zero matrix of height `rows`: the "class-sum to latent" matrix of dz_latent when
// the gradient wrt z already sits in the first L entries of every row
__global__ void gen_identity_kernel(float* __restrict__ m, int rows, int L)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * L) return;
    m[idx] = (idx / L) == (idx % L) ? 1.f : 0.f;
}



// =====================================================================================================================================
// Round 4: fp32-MFMA forms of the stride-1 generic convs (the decoder with the reference's DEFAULT DEC.KERNEL_SIZE 5,
// lib/config/defaults.py:100): exact fp32 products (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 = fmaf chains, 157 TF/s peak), any
// odd kernel size whose weight slice fits LDS, any channel counts and image sizes (bounds are checked per element).  The scalar kernels
// above stay for stride 2 (refinement stack, 3.6 % of the FLOPs) and for slices that do not fit (KERNEL_SIZE 7 with >= 36 channels).
//
// Forward / data gradient (one kernel): implicit GEMM, M = 16 output channels, N = 16 pixels of a tile row, K = 4 channels of one tap.
// A persistent block owns ONE group of 16 output channels and keeps its whole weight slice [tap][reduction channel][16] in LDS
// (KS^2 x C x 64 B: 102 KB for 5 x 5 x 64) for all the tiles it processes - the weights are what a 5 x 5 conv re-reads most; the input
// halo of a 16 x 16 tile is staged per chunk of 16 / 8 / 4 channels as channel-major planes (double-buffered, one barrier per chunk = per
// 400 MFMAs of a wave at 5 x 5 with 16-channel chunks).
//   W(tap, k, n) = wt[tapidx * sT + k * sK + n * sN]:  forward tapidx = tap, (sT, sK, sN) = (Ci Co, Co, 1) on the pack [tap][ci][co];
//   data gradient: the correlation with the flipped kernel, tapidx = KS^2 - 1 - tap, reduction over co, (sT, sK, sN) = (ldi Co, 1, Co).
// =====================================================================================================================================
#ifdef IODINE_TILE_PROF
__device__ unsigned g_gen_prof[TP_MAXBLK * 8];
__device__ unsigned g_genw_prof[TP_MAXBLK * 8];
#endif

template <int KS, int CCH>                                     // CCH: reduction channels per staged chunk (16, 8 or 4: the largest that fits the LDS)
__global__ __launch_bounds__(256)
void gen_conv_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                          const float* __restrict__ aux, float* __restrict__ out, int S, int Ck, int ldin, int Cn, int ldout,
                          int flip, int sT, int sK, int sN, int elu, int ncg, int tiles, int ntiles)
{
    constexpr int KK = KS * KS, PAD = KS / 2, TW = 16 + KS - 1, NPX = TW * TW;
    // Round 5: the staged chunk is CHANNEL-major, one plane of NPXP floats per channel with NPXP = 16 (mod 32): the 16 pixels x 4 channels a
    // wave reads per operand are 16 consecutive words in each of 4 planes that start 16 banks apart - conflict-free (the pixel-major
    // [pixel][8 channels] layout was a 4-way bank conflict on every input operand read, with one wave per SIMD nothing hides it).
    constexpr int NPXP = (NPX + 15) / 32 * 32 + 16;
    constexpr int NH = CCH / 4;                               // MFMA k-steps per tap and chunk (one barrier per chunk)
    constexpr int NQ = CCH / 4;                               // channel quads per pixel: staged with 16-byte loads
    constexpr int NLD = (NPX * NQ + 255) / 256;
    constexpr int RR = 4 + KS - 1;                            // input rows the four tile rows of a wave share
    extern __shared__ __attribute__((aligned(16))) float smem_g[];
    const int Ckp = (Ck + CCH - 1) / CCH * CCH, nchunk = Ckp / CCH;
    float* s_w = smem_g;                                      // [KK][Ckp][16]
    float* s_in = smem_g + (size_t)KK * Ckp * 16;             // [2][CCH][NPXP] + a dump area of 3 NPXP + 1 words (see commit)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // block -> (channel group, tile sequence).  Blocks b, b + 8, b + 16 ... share an XCD: the ncg blocks that need the same input tile at
    // the same time are put on ONE XCD (one L2 fetches the tile) whenever the counts divide; otherwise consecutive blocks take the groups.
    int cg, pb;
    const int nb = gridDim.x / ncg;
    if (nb % 8 == 0) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; cg = j % ncg; pb = (j / ncg) * 8 + x; }
    else { cg = blockIdx.x % ncg; pb = blockIdx.x / ncg; }
    if (pb >= nb) return;                                     // (grid = ncg * nb exactly; defensive)
    TP_DECL;
    // ---- this block's weight slice -> LDS, once ([tap][reduction channel][16]: element e = row * 16 + n, row = tap * Ckp + k) ----
    // (eight loads in flight per thread: a load -> store loop pays the full memory latency a hundred times, 4 % of the kernel)
    {
        const int n = tid & 15, co = cg * 16 + n;
        int k = tid >> 4, tap = 0;
        while (k >= Ckp) { k -= Ckp; ++tap; }
        const int ne = KK * Ckp * 16;
        for (int e0 = tid; e0 < ne; e0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = 0.f;
                if (e0 + 256 * j < ne && k < Ck && co < Cn) v[j] = wt[(size_t)(flip ? KK - 1 - tap : tap) * sT + (size_t)k * sK + (size_t)co * sN];
                k += 16;
#pragma unroll
                for (int f = 0; f < 4; ++f) if (k >= Ckp) { k -= Ckp; ++tap; }       // (Ckp >= 4: at most four rows of the table per step)
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (e0 + 256 * j < ne) s_w[e0 + 256 * j] = v[j];
        }
    }
    const int m = lane & 15, kq = lane >> 4;
    const int co0 = cg * 16 + 4 * kq;                         // the four output channels of this lane (D rows 4 kq + v, column = pixel m)
    float bv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = (bias && co0 + v < Cn) ? bias[co0 + v] : 0.f;
    const bool vec_out = (ldout & 3) == 0 && co0 + 3 < Cn;
    // ---- staging table of this thread, the same for every tile: element i = 16 bytes (channel quad q of halo pixel (py, px)) ----
    // One wave per SIMD: every instruction that is not issued in the shadow of an MFMA is exposed.  So the (pixel, quad) arithmetic is done
    // once per kernel, the bounds once per tile (byte offsets into the slot-image, 0x80000000 = outside: the buffer load returns zeros), and a
    // chunk's loads are one select and one buffer load each, issued BETWEEN the MFMAs of the chunk's first k-step (hand-placed with
    // sched_group_barrier); so are the LDS writes of the staged chunk (last k-step) and the previous tile's epilogue (first k-step).
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    int rel[NLD], pyx[NLD], lw[NLD];
    const int q4 = 4 * (tid % NQ);                            // first channel of this thread's quad within a chunk (256 % NQ == 0: the same for every i)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i, px = e / NQ, py = px / TW, pxx = px % TW;
        rel[i] = ((py * S + pxx) * ldin + q4) * 4;
        pyx[i] = e < NPX * NQ ? (py | pxx << 8) : (255 | 255 << 8);             // (255: never inside)
        lw[i] = e < NPX * NQ ? q4 * NPXP + px : 2 * CCH * NPXP;                  // (threads past the end of the table: the dump area)
    }
    u32x4_ rin[NLD];
    unsigned tvo[NLD];                                         // tile being fetched: byte offsets of the table's elements
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0, 0x00020000);
    const int img_bytes = S * S * ldin * 4;
    auto setup_fetch = [&](int t) {
        const int tx = t % tiles, ty = (t / tiles) % tiles, n = t / (tiles * tiles);
        const int gy0 = ty * 16 - PAD, gx0 = tx * 16 - PAD;
        const bool live = t < ntiles;
        rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)(live ? n : 0) * S * S * ldin), 0, live ? img_bytes : 0, 0x00020000);
        const int tile_off = (gy0 * S + gx0) * ldin * 4;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int gy = gy0 + (pyx[i] & 255), gx = gx0 + (pyx[i] >> 8);
            tvo[i] = ((unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S) ? (unsigned)(tile_off + rel[i]) : 0x80000000u;
        }
    };
    auto fetch = [&](int c) {
        const bool qok = c * CCH + q4 < Ck;                   // (a reduction channel count that is not a multiple of the chunk: zeros)
#pragma unroll
        for (int i = 0; i < NLD; ++i) rin[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(qok ? tvo[i] : 0x80000000u), c * CCH * 4, 0);
    };
    auto commit = [&](int buf) {
        float* dst = s_in + (size_t)buf * CCH * NPXP;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            float* d = dst + lw[i];
            d[0] = __uint_as_float(rin[i].x); d[NPXP] = __uint_as_float(rin[i].y);
            d[2 * NPXP] = __uint_as_float(rin[i].z); d[3 * NPXP] = __uint_as_float(rin[i].w);
        }
    };
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- deferred epilogue: the finished tile's accumulators, store offsets and (data gradient) the activation-derivative operand; the
    // arithmetic and the stores are issued between the MFMAs of the next tile's first k-step (block's first tile: every offset is "outside") ----
    const bool vec_all = (ldout & 3) == 0 && (Cn & 3) == 0;   // 16-byte stores for every lane (else: the plain epilogue at the end of the tile)
    constexpr bool DEFER = NH > 1 && KS < 7;                  // (7 x 7: the two operand sets leave no registers for it - plain order, no spills)
    f32x4 res[4];
    u32x4_ ax[4];
    unsigned evo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { res[r] = f32x4{0.f, 0.f, 0.f, 0.f}; ax[r] = u32x4_{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u}; evo[r] = 0x80000000u; }
    __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);
    const int out_bytes = S * S * ldout * 4;
    auto epilogue_math = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = res[r][j] + bv[j];
                const float ev = elu1_fast(v[j]);
                v[j] = elu ? ev : v[j];
                v[j] *= elu1_grad_from_out(__uint_as_float(ax[r][j]));
            }
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                   rs_out, (int)evo[r], 0, 0);
        }
    };
    // plain epilogue (channel counts / strides without 16-byte stores, and the kernels with one k-step per chunk)
    auto epilogue_now = [&](int t) {
        const int tx = t % tiles, ty = (t / tiles) % tiles, n = t / (tiles * tiles);
        const int x = tx * 16 + m;
        if (x < S && co0 < Cn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = ty * 16 + 4 * wv + r;
                if (y >= S) continue;
                const size_t o = (((size_t)n * S + y) * S + x) * ldout + co0;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = acc[r][j] + bv[j]; if (elu) v[j] = elu1_fast(v[j]); }
                if (vec_out) {
                    if (aux) {
                        const float4 a = *reinterpret_cast<const float4*>(aux + o);
                        v[0] *= elu1_grad_from_out(a.x); v[1] *= elu1_grad_from_out(a.y);
                        v[2] *= elu1_grad_from_out(a.z); v[3] *= elu1_grad_from_out(a.w);
                    }
                    *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (co0 + j >= Cn) continue;
                        if (aux) v[j] *= elu1_grad_from_out(aux[o + j]);
                        out[o + j] = v[j];
                    }
                }
            }
        }
    };
    // end of a tile on the deferred path: accumulators -> res, store offsets, the derivative operand's loads (issued here, consumed an item later)
    auto retire_tile = [&](int t) {
        const int tx = t % tiles, ty = (t / tiles) % tiles, n = t / (tiles * tiles);
        const int x = tx * 16 + m;
        rs_out = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)n * S * S * ldout, 0, out_bytes, 0x00020000);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = ty * 16 + 4 * wv + r;
            evo[r] = (x < S && y < S && co0 < Cn) ? (unsigned)(((y * S + x) * ldout + co0) * 4) : 0x80000000u;
            res[r] = acc[r];
            acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (aux) {                                            // (uniform)
            const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aux + (size_t)n * S * S * ldout), 0, out_bytes, 0x00020000);
#pragma unroll
            for (int r = 0; r < 4; ++r) ax[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_aux, (int)evo[r], 0, 0);
        }
    };
    int t = pb, c = 0, buf = 0;
    TP_STAMP(0);
    setup_fetch(t);
    fetch(0);
    commit(0);
    __syncthreads();                                          // weights and the first chunk staged
    TP_STAMP(1);
    // one flat sequence of (tile, chunk) items: the next item's chunk - of the NEXT tile after a tile's last chunk - is in flight under this
    // item's MFMAs and is written to the other LDS buffer between the MFMAs of the item's last k-step
    while (t < ntiles) {
        const bool last_chunk = c + 1 == nchunk;
        const int tn = last_chunk ? t + nb : t, cn = last_chunk ? 0 : c + 1;
        if (last_chunk) setup_fetch(tn);                      // (past the last tile: an empty buffer, every load returns zeros)
        if constexpr (!DEFER) fetch(cn);
        TP_STAMP(2);
        const float* si = s_in + (size_t)buf * CCH * NPXP + (size_t)kq * NPXP + (4 * wv) * TW + m;
        const float* sw = s_w + (size_t)(c * CCH + kq) * 16 + m;
        // All operands of a k-step (4 channels) are register-resident before its 4 KK MFMAs issue back to back: the KK weight values of
        // this lane and the (4 + KS - 1) x KS input values its four tile rows share between their taps (40 + 25 LDS reads at KS = 5).
        // Two operand sets: the reads of k-step h + 1 are issued between the MFMAs of k-step h, so that only the first k-step after a
        // barrier waits for the LDS.
        auto ld = [&](int hh, float (&bw)[KK], float (&av)[RR][KS]) {
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) bw[tap] = sw[(size_t)tap * Ckp * 16 + hh * 64];
#pragma unroll
            for (int rr = 0; rr < RR; ++rr)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) av[rr][kx] = si[hh * 4 * NPXP + rr * TW + kx];
        };
        auto mm = [&](const float (&bw)[KK], const float (&av)[RR][KS]) {
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                const int ky = tap / KS, kx = tap % KS;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[tap], av[r + ky][kx], acc[r], 0, 0, 0);
            }
        };
        // (hipcc's scheduler, left alone, sinks every read to just in front of its first use - a full LDS latency in front of every few
        // MFMAs; the group barriers pin "one MFMA, one LDS access / memory instruction / a few VALU instructions")
        auto kstep = [&](auto hh_c, auto epi_c, float (&bw)[KK], float (&av)[RR][KS], float (&bwn)[KK], float (&avn)[RR][KS]) {
            constexpr int hh = decltype(hh_c)::value;
            constexpr bool EPI = decltype(epi_c)::value;      // first k-step of a tile's first chunk: the previous tile's epilogue rides along
            constexpr int NRD = KK + RR * KS;
            if constexpr (hh + 1 < NH) {
                ld(hh + 1, bwn, avn);
                if constexpr (hh == 0 && DEFER) fetch(cn);
                if constexpr (EPI) epilogue_math();
                mm(bw, av);
                // per MFMA (32 cycles = 8 issue slots): one LDS read of the next k-step, then the chunk's loads, and at most 3 VALU instructions
#pragma unroll
                for (int i = 0; i < 4 * KK; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (hh == 0 && DEFER && i >= 8 && i < 8 + NLD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (hh == 0 && DEFER) __builtin_amdgcn_sched_group_barrier(0x002, EPI ? 3 : 1, 0);
                    if (EPI && i >= 4 * KK - 4) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                }
            } else {
                commit(buf ^ 1);                              // (no next item: zeros into a buffer nobody reads again)
                mm(bw, av);
#pragma unroll
                for (int i = 0; i < 4 * NLD; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        float bwA[KK], avA[RR][KS], bwB[KK], avB[RR][KS];
        ld(0, bwA, avA);
        __builtin_amdgcn_sched_barrier(0);
        if (DEFER && c == 0) kstep(std::integral_constant<int, 0>{}, std::true_type{}, bwA, avA, bwB, avB);
        else kstep(std::integral_constant<int, 0>{}, std::false_type{}, bwA, avA, bwB, avB);
        if constexpr (NH > 1) kstep(std::integral_constant<int, 1>{}, std::false_type{}, bwB, avB, bwA, avA);
        if constexpr (NH > 2) {
            kstep(std::integral_constant<int, 2>{}, std::false_type{}, bwA, avA, bwB, avB);
            kstep(std::integral_constant<int, 3>{}, std::false_type{}, bwB, avB, bwA, avA);
        }
        TP_STAMP(3);
        __syncthreads();
        TP_STAMP(5);
        if (last_chunk) {
            if (DEFER && vec_all) retire_tile(t);
            else {
                epilogue_now(t);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            TP_STAMP(6);
        }
        t = tn; c = cn; buf ^= 1;
    }
    if constexpr (DEFER) epilogue_math();                      // the block's last tile
    TP_FLUSH(g_gen_prof);
}

// Weight gradient, stride 1: dW[tap][ci][co] = sum_px in[px + tap][ci] * dout[px][co] as D[32 ci x 32 co] += A[32 x 2 px] B[2 px x 32]
// (v_mfma_f32_32x32x2_f32); a block owns ONE tap and one of GEN_WGRAD_SLICES row ranges, its four waves the (ci tile, co tile) pairs;
// both operands are read straight from global memory (4-byte loads, coalesced over the channels, sixteen in flight per lane).  Partial
// tiles in the layout of gen_conv_wgrad_reduce_kernel; the bias partial comes from the blocks of tap 0.
template <int KS>
__global__ __launch_bounds__(256)
void gen_wgrad_mfma_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ part, int N, int S, int Ci,
                           int ldc, int Co, int nslice)
{
    constexpr int KK = KS * KS, PAD = KS / 2;
    const int tap = blockIdx.x % KK, slice = (blockIdx.x / KK) % nslice, pgrp = blockIdx.x / (KK * nslice);
    const int ky = tap / KS, kx = tap % KS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int nct = (Co + 31) / 32, npair = ((Ci + 31) / 32) * nct;
    const int pair = pgrp * 4 + wv;
    const int per = KK * Ci * Co + Co;
    if (pair >= npair) return;
    const int cit = pair / nct, cot = pair % nct;
    const int ci = cit * 32 + li, co = cot * 32 + li;
    const bool civ = ci < Ci, cov = co < Co;
    const long long R = (long long)N * S;
    const long long r0 = R * slice / nslice, r1 = R * (slice + 1) / nslice;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float bsum = 0.f;
    for (long long row = r0; row < r1; ++row) {
        const int y = (int)(row % S);
        const long long n = row / S;
        const int iy = y + ky - PAD;
        const bool rowv = (unsigned)iy < (unsigned)S;
        const float* ip = in + ((size_t)n * S + (rowv ? iy : 0)) * S * (size_t)ldc + ci;
        const float* dp = dout + ((size_t)n * S + y) * S * (size_t)Co + co;
        for (int x0 = 0; x0 < S; x0 += 16) {
            float a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = x0 + 2 * j + kh, ix = x + kx - PAD;
                a[j] = (civ && rowv && x < S && (unsigned)ix < (unsigned)S) ? ip[(size_t)ix * ldc] : 0.f;
                b[j] = (cov && x < S) ? dp[(size_t)x * Co] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
                bsum += b[j];
            }
        }
    }
    float* pw = part + (size_t)slice * per;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int cr = cit * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
        if (cr < Ci && cov) pw[((size_t)tap * Ci + cr) * Co + co] = acc[q];
    }
    if (tap == 0 && cit == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (kh == 0 && cov) pw[(size_t)KK * Ci * Co + co] = bsum;
    }
}

constexpr int GEN_ROWS_MAXV = 16, GEN_ROWS_U = 4;
// geometry of gen_wgrad_rows_kernel's staged rows: operand blocks of GEN_ROWS_U pixel pairs, an even number of them per row
__host__ __device__ inline int gen_rows_nblk(int S) { const int nb = ((S + 1) / 2 + GEN_ROWS_U - 1) / GEN_ROWS_U; return (nb + 1) & ~1; }
__host__ __device__ inline int gen_rows_wp(int S, int k) { const int w = 2 * GEN_ROWS_U * gen_rows_nblk(S) + k - 1 + 1; return w > S + 2 * (k / 2) + 1 ? w : S + 2 * (k / 2) + 1; }
__host__ __device__ inline int gen_rows_sd(int S) { const int w = 2 * GEN_ROWS_U * gen_rows_nblk(S) + 1; return w > S + 1 ? w : S + 1; }

// Round 5: the same weight gradient with the operands staged ONCE per kernel ROW.  The kernel above gives every tap its own blocks, each
// streaming both tensors from global memory again (KS^2 = 25 passes over `in` and `dout` at KS = 5: 5.9 GB of L2 / HBM traffic per launch at
// the CLEVR shapes with 28 slot-images, and the load -> 8 MFMAs -> load chain hides its latency only through occupancy: 34 % of the fp32 matrix
// peak).  Here a block owns one kernel row ky and one row slice; per image row it stages the input row y + ky - PAD (S + 2 PAD pixels, zero
// margins) and the gradient row y as planes of 32 channels in LDS - the next row is fetched into registers while the current one is multiplied -
// and its four waves ((ci tile, co tile) pairs) issue the KS taps kx of that row from it: KS accumulators of 32 x 32 per wave, 5 MFMAs per two
// LDS reads.  Channel counts and row strides that are multiples of 4 (float4 staging), rows that fit the LDS; everything else keeps the kernel above.
template <int KS>
__global__ __launch_bounds__(256, 2)
void gen_wgrad_rows_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ part, int N, int S, int Ci,
                           int ldc, int Co, int nslice)
{
    constexpr int KK = KS * KS, PAD = KS / 2;
    constexpr int MAXV = GEN_ROWS_MAXV;                        // 16-byte loads per thread and staged row pair (launcher: fits)
    constexpr int U = GEN_ROWS_U;                              // pixel pairs per operand block: 2 U + KS - 1 input values + U gradient values feed U KS MFMAs
    constexpr int NAW = 2 * U + KS - 1;
    extern __shared__ __attribute__((aligned(16))) float smem_gw[];
    const int ky = blockIdx.x % KS, slice = (blockIdx.x / KS) % nslice, pgrp = blockIdx.x / (KS * nslice);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int nit = (Ci + 31) / 32, nct = (Co + 31) / 32, npair = nit * nct;
    const int nblk = gen_rows_nblk(S);                         // operand blocks per row (even: two register sets alternate)
    const int Wp = gen_rows_wp(S, KS), Sd = gen_rows_sd(S);    // staged pixels per row (zero margins; the last block may run past the row)
    float* s_a = smem_gw;                                      // [nit][Wp][32]
    float* s_d = smem_gw + (size_t)nit * Wp * 32;              // [nct][Sd][32], then a 16-byte dump slot
    const int dump = (nit * Wp + nct * Sd) * 32;
    const int pair = pgrp * 4 + wv;
    const bool wave_on = pair < npair;
    const int cit = wave_on ? pair / nct : 0, cot = wave_on ? pair % nct : 0;
    const int co = cot * 32 + li;
    const bool cov = co < Co;
    const int per = KK * Ci * Co + Co;
    const long long R = (long long)N * S;
    const long long r0 = R * slice / nslice, r1 = R * (slice + 1) / nslice;
    // zero the planes once: channel pads, pixel margins and the pixels past the row stay zero (staging only writes real elements)
    for (int e = tid; e < dump + 4; e += 256) smem_gw[e] = 0.f;

    // staging table of this thread (the same for every row): entries [0, kA) are 16-byte pieces of the input row, [kA, kA + kD) of the gradient
    // row; byte offset within the row (0x80000000 = none: the buffer load returns zeros) and LDS word offset (none: the dump slot).  A row is then
    // MAXV buffer loads without any address arithmetic, and a row outside the image an empty buffer.
    const int A4 = (Ci + 3) / 4, D4 = Co / 4;                 // 16-byte pieces per pixel (Ci rounded up: the row stride covers the pad channels)
    const int na = S * A4, nd = S * D4, kA = (na + 255) / 256;
    unsigned voff[MAXV];
    int loff[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        voff[k] = 0x80000000u; loff[k] = dump;
        if (k < kA) {
            const int e = tid + 256 * k, x = e / A4, c4 = e % A4;
            if (e < na) { voff[k] = (unsigned)((x * ldc + c4 * 4) * 4); loff[k] = ((c4 >> 3) * Wp + x + PAD) * 32 + (c4 & 7) * 4; }
        } else {
            const int f = tid + 256 * (k - kA), x = f / D4, c4 = f % D4;
            if (f < nd) { voff[k] = (unsigned)((x * Co + c4 * 4) * 4); loff[k] = nit * Wp * 32 + ((c4 >> 3) * Sd + x) * 32 + (c4 & 7) * 4; }
        }
    }
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    u32x4_ rv[MAXV];
    auto fetch = [&](long long row) {
        const int y = (int)(row % S);
        const long long n = row / S;
        const int iy = y + ky - PAD;
        const bool rowv = (unsigned)iy < (unsigned)S;
        const float* ip = in + ((size_t)n * S + (rowv ? iy : 0)) * S * (size_t)ldc;
        const float* dp = dout + ((size_t)n * S + y) * S * (size_t)Co;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ip), 0, rowv ? S * ldc * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dp), 0, S * Co * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            if (k < kA) rv[k] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff[k], 0, 0);    // (uniform branch)
            else rv[k] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)voff[k], 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < MAXV; ++k) *reinterpret_cast<u32x4_*>(smem_gw + loff[k]) = rv[k];
    };
    f32x16 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    float bsum = 0.f;
    const bool do_bias = ky == 0 && cit == 0 && wave_on;
    const float* ap = s_a + ((size_t)cit * Wp + kh) * 32 + li;
    const float* dpl = s_d + ((size_t)cot * Sd + kh) * 32 + li;
    // operands of block bq: gradient values of its U pixel pairs, input values of the 2 U + KS - 1 pixels they touch (tap kx of pair u: aw[2 u + kx])
    auto ld = [&](int bq, float (&aw)[NAW], float (&bw)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) bw[u] = dpl[(size_t)(2 * U * bq + 2 * u) * 32];
#pragma unroll
        for (int j = 0; j < NAW; ++j) aw[j] = ap[(size_t)(2 * U * bq + j) * 32];
    };
    auto mm = [&](const float (&aw)[NAW], const float (&bw)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[2 * u + kx], bw[u], acc[kx], 0, 0, 0);
            bsum += bw[u];
        }
    };
    // (the group barriers pin "one MFMA of this block, one LDS read of the next": left alone, hipcc puts every read right in front of its use)
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < NAW + U; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    TP_DECL;
    if (r0 < r1) fetch(r0);
    __syncthreads();                                           // planes zeroed
    TP_STAMP(0);
    for (long long row = r0; row < r1; ++row) {
        commit();
        TP_STAMP(1);
        __syncthreads();
        TP_STAMP(2);
        if (row + 1 < r1) fetch(row + 1);                      // in flight under this row's MFMAs
        TP_STAMP(3);
        const int iy = (int)(row % S) + ky - PAD;
        if (wave_on && ((unsigned)iy < (unsigned)S || do_bias)) {
            float awA[NAW], bwA[U], awB[NAW], bwB[U];
            ld(0, awA, bwA);
            __builtin_amdgcn_sched_barrier(0);
            for (int bq = 0; bq < nblk; bq += 2) {
                ld(bq + 1, awB, bwB);
                mm(awA, bwA);
                interleave();
                ld(bq + 2, awA, bwA);                          // (past the last block: never multiplied; LDS reads past the allocation return 0)
                mm(awB, bwB);
                interleave();
            }
        }
        TP_STAMP(4);
        __syncthreads();                                       // every wave is done with the planes
        TP_STAMP(5);
    }
    TP_FLUSH(g_genw_prof);
    if (!wave_on) return;
    float* pw = part + (size_t)slice * per;
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cr = cit * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (cr < Ci && cov) pw[((size_t)(ky * KS + kx) * Ci + cr) * Co + co] = acc[kx][q];
        }
    if (do_bias) {
        bsum += __shfl_xor(bsum, 32);
        if (kh == 0 && cov) pw[(size_t)KK * Ci * Co + co] = bsum;
    }
}

// Round 5: weight gradient of a conv with FOUR output channels (the decoder's output conv, iodine.py:422,435) in GEMM form:
//     dW[tap][ci][co] = sum_q in[q][ci] * g[q - (tap - centre)][co]        rows ci, columns j = tap * 4 + co (4 KS^2 of them), K = pixels
// The tap shift sits on the 4-channel gradient g - a KS-row window with zero margins, 10 KB of LDS at S = 128 - and the wide operand needs no
// halo: 4 KS^2 / 32 column tiles x Ci / 32 row tiles = 8 MFMAs (v_mfma_f32_32x32x2_f32) per pixel pair at 5 x 5 x 64, where the row-staged
// kernel above, with its 4 output channels padded to a 32-wide tile, issues 50 (0.81 ms per launch at the CLEVR shapes for 6 GFLOP).
// A block owns a slice of image rows; per row it stages the input row (planes of 32 channels) and the KS gradient rows around it, the next row in
// flight in registers; its four waves share the (row tile, column tile) pairs, NP per wave.  Partial tiles in the layout of
// gen_conv_wgrad_reduce_kernel; the bias partial comes from the centre tap's columns (g itself).
template <int KS, int NP>
__global__ __launch_bounds__(256, 2)
void gen_wgrad_out_kernel(const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ part, int N, int S, int Ci, int ldc,
                          int nslice)
{
    constexpr int KK = KS * KS, PAD = KS / 2, NJ = 4 * KK, NCT = (NJ + 31) / 32, Co = 4;
    constexpr int MAXV = 16;
    extern __shared__ __attribute__((aligned(16))) float smem_go[];
    const int slice = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int nit = (Ci + 31) / 32, npair = nit * NCT;
    const int Wa = S + 1, Wg = S + 2 * PAD + 2;
    float* s_a = smem_go;                                      // [nit][Wa][32]
    float* s_g = smem_go + (size_t)nit * Wa * 32;              // [KS][Wg][4]
    const int per = KK * Ci * Co + Co;
    const long long R = (long long)N * S;
    const long long r0 = R * slice / nslice, r1 = R * (slice + 1) / nslice;
    for (int e = tid; e < nit * Wa * 32 + KS * Wg * 4; e += 256) smem_go[e] = 0.f;

    const int A4 = Ci / 4;
    const int na = S * A4, ng = KS * S;
    float4 rv[MAXV];
    auto fetch = [&](long long row) {
        const int y = (int)(row % S);
        const long long n = row / S;
        const float* ip = in + ((size_t)n * S + y) * S * (size_t)ldc;
        const float* gp = g + (size_t)n * S * S * 4;
        int tv = tid;
        asm volatile("" : "+v"(tv));                           // (element arithmetic recomputed per row: hoisted it spills)
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int e = tv + 256 * k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < na) v = *reinterpret_cast<const float4*>(ip + (size_t)(e / A4) * ldc + (e % A4) * 4);
            else if (e < na + ng) {
                const int f = e - na, r = f / S, x = f - r * S, yy = y - PAD + r;
                if ((unsigned)yy < (unsigned)S) v = *reinterpret_cast<const float4*>(gp + ((size_t)yy * S + x) * 4);
            }
            rv[k] = v;
        }
    };
    auto commit = [&]() {
        int tv = tid;
        asm volatile("" : "+v"(tv));
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int e = tv + 256 * k;
            if (e < na) { const int x = e / A4, c4 = e % A4; *reinterpret_cast<float4*>(s_a + ((size_t)(c4 >> 3) * Wa + x) * 32 + (c4 & 7) * 4) = rv[k]; }
            else if (e < na + ng) { const int f = e - na, r = f / S, x = f - r * S; *reinterpret_cast<float4*>(s_g + ((size_t)r * Wg + x + PAD) * 4) = rv[k]; }
        }
    };
    // the pairs of this wave: p = wv + 4 i -> (row tile cit, column tile ct); column j = ct * 32 + li = tap * 4 + co reads the gradient at
    // staged row KS - 1 - ky, staged column x + 2 PAD - kx (columns past 4 KS^2 read tap 0 and are never written)
    int aoff[NP], boff[NP];
    bool pon[NP], jok[NP];
    f32x16 acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = wv + 4 * i;
        pon[i] = p < npair;
        const int cit = pon[i] ? p / NCT : 0, ct = pon[i] ? p % NCT : 0, j = ct * 32 + li;
        jok[i] = j < NJ;
        const int tap = jok[i] ? j >> 2 : 0, ky = tap / KS, kx = tap % KS;
        aoff[i] = (cit * Wa + kh) * 32 + li;
        boff[i] = ((KS - 1 - ky) * Wg + 2 * PAD - kx + kh) * 4 + (j & 3);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    }
    // bias: the wave that owns (row tile 0, the column tile of the centre tap)
    constexpr int JC = 4 * (KK / 2);
    const bool bias_wave = wv == (JC / 32) % 4;
    constexpr int BI = (JC / 32) / 4;                          // index of that pair in its wave
    const bool bias_lane = bias_wave && (li >> 2) == (JC % 32) / 4;
    float bsum = 0.f;
    if (r0 < r1) fetch(r0);
    __syncthreads();                                           // planes zeroed
    for (long long row = r0; row < r1; ++row) {
        commit();
        __syncthreads();
        if (row + 1 < r1) fetch(row + 1);                      // in flight under this row's MFMAs
        for (int x0 = 0; x0 < S; x0 += 2) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (!pon[i]) continue;                         // (uniform per wave)
                const float a = s_a[aoff[i] + x0 * 32], b = s_g[boff[i] + x0 * 4];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                if (i == BI && bias_lane) bsum += b;
            }
        }
        __syncthreads();                                       // every wave is done with the planes
    }
    float* pw = part + (size_t)slice * per;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (!pon[i] || !jok[i]) continue;
        const int p = wv + 4 * i, cit = p / NCT, j = (p % NCT) * 32 + li;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cr = cit * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (cr < Ci) pw[((size_t)(j >> 2) * Ci + cr) * Co + (j & 3)] = acc[i][q];
        }
    }
    // (the pair (row tile 0, centre column tile) is pair index JC / 32 < 4 NP: row tile 0 comes first)
    bsum += __shfl_xor(bsum, 32);
    if (bias_lane && kh == 0) pw[(size_t)KK * Ci * Co + (li & 3)] = bsum;
}

// LDS bytes of gen_wgrad_out_kernel, 0 = does not qualify (four output channels, 16-byte staging, <= 16 float4 per thread and row, <= 80 KB)
inline size_t gen_wgrad_out_lds(int S, int Ci, int ldc, int Co, int k)
{
    if (Co != 4 || (Ci & 3) || (ldc & 3)) return 0;
    if ((size_t)S * (Ci / 4 + k) > (size_t)16 * 256) return 0;
    const int npair = ((Ci + 31) / 32) * ((4 * k * k + 31) / 32);
    if (npair > 32) return 0;                                  // (at most 8 pairs per wave)
    const size_t b = ((size_t)((Ci + 31) / 32) * (S + 1) * 32 + (size_t)k * (S + 2 * (k / 2) + 2) * 4) * sizeof(float);
    return b <= 80 * 1024 ? b : 0;
}

// LDS bytes of gen_wgrad_rows_kernel, 0 = the shape does not qualify (float4 staging, at most 16 float4 per thread and row, <= 80 KB: two blocks per CU)
inline size_t gen_wgrad_rows_lds(int S, int Ci, int ldc, int Co, int k)
{
    if (Co % 4 != 0 || ldc % 4 != 0 || ((Ci + 3) & ~3) > ldc) return 0;
    const int na = S * ((Ci + 3) / 4), nd = S * (Co / 4);
    if ((na + 255) / 256 + (nd + 255) / 256 > GEN_ROWS_MAXV) return 0;
    if ((size_t)S * ldc * 4 >= ((size_t)1 << 31) || (size_t)S * Co * 4 >= ((size_t)1 << 31)) return 0;   // (32-bit buffer offsets)
    const size_t b = ((size_t)((Ci + 31) / 32) * gen_rows_wp(S, k) + (size_t)((Co + 31) / 32) * gen_rows_sd(S)) * 32 * sizeof(float) + 16;
    return b <= 80 * 1024 ? b : 0;
}

inline unsigned gen_blocks(size_t total) { return (unsigned)((total + 255) / 256); }

// LDS bytes of the MFMA form with chunks of cch channels: weight slice + two channel-major halo chunks; 0 = does not apply (stride 2, or the
// slice does not fit)
inline size_t gen_mfma_lds_cch(int k, int Ck, int s, int cch)
{
    if (s != 1 || (k != 3 && k != 5 && k != 7)) return 0;
    const int Ckp = (Ck + cch - 1) / cch * cch, TW = 16 + k - 1, NPXP = (TW * TW + 15) / 32 * 32 + 16;
    const size_t b = ((size_t)k * k * Ckp * 16 + (size_t)2 * NPXP * cch + 3 * NPXP + 4) * sizeof(float);
    return b <= 160 * 1024 ? b : 0;
}
// chunk width: the smallest of 4 / 8 / 16 that covers the reduction channels, stepping down while the LDS does not fit; 0 = the scalar kernels
inline int gen_mfma_cch(int k, int Ck, int s)
{
    int cch = Ck <= 4 ? 4 : (Ck <= 8 || k == 7) ? 8 : 16;    // (7 x 7: two operand sets of 119 registers - 8-channel chunks keep it free of spills)
    while (cch >= 4 && !gen_mfma_lds_cch(k, Ck, s, cch)) cch >>= 1;
    return cch >= 4 ? cch : 0;
}
// (16-byte staging: channel count and channel stride of the input must be multiples of 4)
inline size_t gen_mfma_lds(int k, int Ck, int ldin, int s)
{
    if ((Ck & 3) || (ldin & 3)) return 0;
    const int cch = gen_mfma_cch(k, Ck, s);
    return cch ? gen_mfma_lds_cch(k, Ck, s, cch) : 0;
}

inline int npair_out(int Ci, int k) { return ((Ci + 31) / 32) * ((4 * k * k + 31) / 32); }

template <int KS, int NP>
hipError_t gen_wgrad_out_launch_np(hipStream_t st, const float* in, const float* g, float* part, int N, int S, int Ci, int ldc, int nsl, size_t lds)
{
    static std::atomic<unsigned> attr_devs{0};
    if (hipError_t e = iod_set_max_lds((const void*)gen_wgrad_out_kernel<KS, NP>, 80 * 1024, attr_devs); e != hipSuccess) return e;
    hipLaunchKernelGGL((gen_wgrad_out_kernel<KS, NP>), dim3(nsl), dim3(256), lds, st, in, g, part, N, S, Ci, ldc, nsl);
    return hipGetLastError();
}
template <int KS>
hipError_t gen_wgrad_out_launch_ks(hipStream_t st, const float* in, const float* g, float* part, int N, int S, int Ci, int ldc, int nsl, int np4,
                                   size_t lds)
{
    if (np4 <= 1) return gen_wgrad_out_launch_np<KS, 1>(st, in, g, part, N, S, Ci, ldc, nsl, lds);
    if (np4 <= 2) return gen_wgrad_out_launch_np<KS, 2>(st, in, g, part, N, S, Ci, ldc, nsl, lds);
    if (np4 <= 4) return gen_wgrad_out_launch_np<KS, 4>(st, in, g, part, N, S, Ci, ldc, nsl, lds);
    return gen_wgrad_out_launch_np<KS, 8>(st, in, g, part, N, S, Ci, ldc, nsl, lds);
}
inline hipError_t gen_wgrad_out_launch(hipStream_t st, const float* in, const float* g, float* part, int N, int S, int Ci, int ldc, int k, int nsl,
                                       int np4, size_t lds)
{
    if (k == 3) return gen_wgrad_out_launch_ks<3>(st, in, g, part, N, S, Ci, ldc, nsl, np4, lds);
    if (k == 5) return gen_wgrad_out_launch_ks<5>(st, in, g, part, N, S, Ci, ldc, nsl, np4, lds);
    return gen_wgrad_out_launch_ks<7>(st, in, g, part, N, S, Ci, ldc, nsl, np4, lds);
}

template <int KS, int CCH>
hipError_t gen_mfma_launch_cch(hipStream_t st, const float* in, const float* wt, const float* bias, const float* aux, float* out, int N, int S,
                               int Ck, int ldin, int Cn, int ldout, int flip, int sT, int sK, int sN, int elu, size_t lds)
{
    static std::atomic<unsigned> attr_devs{0};
    if (hipError_t e = iod_set_max_lds((const void*)gen_conv_mfma_kernel<KS, CCH>, 160 * 1024, attr_devs); e != hipSuccess) return e;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    const int ncg = (Cn + 15) / 16, tiles = (S + 15) / 16, ntiles = N * tiles * tiles;
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int nb = std::max(1, std::min(ntiles, per_cu * n_cu / ncg));
    hipLaunchKernelGGL((gen_conv_mfma_kernel<KS, CCH>), dim3(ncg * nb), dim3(256), lds, st, in, wt, bias, aux, out, S, Ck, ldin, Cn, ldout, flip,
                       sT, sK, sN, elu, ncg, tiles, ntiles);
#ifdef IODINE_TILE_PROF
    if (getenv("IODINE_GEN_PROF")) {
        const int nbk = std::min(ncg * nb, TP_MAXBLK);
        std::vector<unsigned> hp((size_t)nbk * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_gen_prof), hp.size() * sizeof(unsigned));
        static const char* names[8] = {"weights/loop", "tile prologue", "fetch issue", "k-steps", "commit", "barrier", "epilogue", "-"};
        double sum[8] = {0}, tot = 0;
        for (int b2 = 0; b2 < nbk; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[(size_t)b2 * 8 + i];
        for (int i = 0; i < 8; ++i) tot += sum[i] / nbk;
        fprintf(stderr, "[gen prof k%d cch%d Ck%d Cn%d] ticks per block (%d tiles), total %.0f:", KS, CCH, Ck, Cn, (ntiles + nb - 1) / nb, tot);
        for (int i = 0; i < 7; ++i) fprintf(stderr, " %s %.0f |", names[i], sum[i] / nbk);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

template <int KS>
hipError_t gen_mfma_launch(hipStream_t st, const float* in, const float* wt, const float* bias, const float* aux, float* out, int N, int S,
                           int Ck, int ldin, int Cn, int ldout, int flip, int sT, int sK, int sN, int elu, size_t lds)
{
    switch (gen_mfma_cch(KS, Ck, 1)) {
    case 16:
        if constexpr (KS < 7) return gen_mfma_launch_cch<KS, 16>(st, in, wt, bias, aux, out, N, S, Ck, ldin, Cn, ldout, flip, sT, sK, sN, elu, lds);
        return hipErrorInvalidValue;
    case 8: return gen_mfma_launch_cch<KS, 8>(st, in, wt, bias, aux, out, N, S, Ck, ldin, Cn, ldout, flip, sT, sK, sN, elu, lds);
    default: return gen_mfma_launch_cch<KS, 4>(st, in, wt, bias, aux, out, N, S, Ck, ldin, Cn, ldout, flip, sT, sK, sN, elu, lds);
    }
}


}  // namespace

hipError_t launch_gen_pack_weights(hipStream_t st, const float* w, int Co, int Ci, int k, float* wt)
{
    hipLaunchKernelGGL(gen_pack_weights_kernel, dim3(gen_blocks((size_t)Co * Ci * k * k)), dim3(256), 0, st, w, Co, Ci, k * k, wt);
    return hipGetLastError();
}

hipError_t launch_gen_conv_fwd(hipStream_t st, const float* in, const float* wt, const float* bias, float* out, int N, int Si, int Ci,
                               int ldc, int Co, int k, int s, int elu, unsigned chmask)
{
    if (const size_t lds = gen_mfma_lds(k, Ci, ldc, s)) {
        // [tap][ci][co] pack: W(tap, k = ci, n = co)
        if (k == 3) return gen_mfma_launch<3>(st, in, wt, bias, nullptr, out, N, Si, Ci, ldc, Co, Co, 0, Ci * Co, Co, 1, elu, lds);
        if (k == 5) return gen_mfma_launch<5>(st, in, wt, bias, nullptr, out, N, Si, Ci, ldc, Co, Co, 0, Ci * Co, Co, 1, elu, lds);
        return gen_mfma_launch<7>(st, in, wt, bias, nullptr, out, N, Si, Ci, ldc, Co, Co, 0, Ci * Co, Co, 1, elu, lds);
    }
    if (s == 2 && gen_s2_fwd_ok(k, Ci, ldc, Co))         // stride 2 on the matrix pipe (kernels_gens2.hip, round 5)
        return launch_gen_s2_fwd(st, in, wt, bias, out, N, Si, Ci, ldc, Co, k, elu, chmask);
    const int So = (Si - 1) / s + 1;
    const size_t total = (size_t)N * So * So * Co;
    hipLaunchKernelGGL(gen_conv_fwd_kernel, dim3(gen_blocks(total)), dim3(256), 0, st, in, wt, bias, out, Si, So, Ci, ldc, Co, k, s, elu, total);
    return hipGetLastError();
}

hipError_t launch_gen_conv_dgrad(hipStream_t st, const float* dout, const float* wt, const float* aux, float* din, int N, int Si, int Ci,
                                 int ldi, int Co, int k, int s)
{
    if (const size_t lds = gen_mfma_lds(k, Co, Co, s)) {
        // correlation of dout with the flipped kernel: reduction over co, W(tap, k = co, n = ci) = wt[(KK - 1 - tap)][ci][co]
        if (k == 3) return gen_mfma_launch<3>(st, dout, wt, nullptr, aux, din, N, Si, Co, Co, Ci, ldi, 1, ldi * Co, 1, Co, 0, lds);
        if (k == 5) return gen_mfma_launch<5>(st, dout, wt, nullptr, aux, din, N, Si, Co, Co, Ci, ldi, 1, ldi * Co, 1, Co, 0, lds);
        return gen_mfma_launch<7>(st, dout, wt, nullptr, aux, din, N, Si, Co, Co, Ci, ldi, 1, ldi * Co, 1, Co, 0, lds);
    }
    if (s == 2 && gen_s2_dgrad_ok(k, Ci, ldi, Co))
        return launch_gen_s2_dgrad(st, dout, wt, aux, din, N, Si, Ci, ldi, Co, k);
    const int So = (Si - 1) / s + 1;
    const size_t total = (size_t)N * Si * Si * Ci;
    hipLaunchKernelGGL(gen_conv_dgrad_kernel, dim3(gen_blocks(total)), dim3(256), 0, st, dout, wt, aux, din, Si, So, Ci, ldi, Co, k, s, total);
    return hipGetLastError();
}

// (the GEMM form of a 4-output-channel conv takes one slice per resident block: up to GEN_WGRAD_OUT_SLICES_MAX small partials)
size_t gen_wgrad_scratch_floats(int Ci, int Co, int k)
{
    return (size_t)(Co == 4 ? GEN_WGRAD_OUT_SLICES_MAX : GEN_WGRAD_SLICES_MAX) * ((size_t)k * k * Ci * Co + Co);
}

hipError_t launch_gen_conv_wgrad(hipStream_t st, const float* in, const float* dout, float* scratch, int N, int Si, int Ci, int ldc,
                                 int Ci_dst, int Co, int k, int s, float alpha, float* gw, float* gb, unsigned chmask)
{
    const int So = (Si - 1) / s + 1;
    const size_t per = (size_t)k * k * Ci * Co + Co;
    if (s == 1 && (k == 3 || k == 5 || k == 7)) {
        const int npair = ((Ci + 31) / 32) * ((Co + 31) / 32), ngrp = (npair + 3) / 4;
        if (const size_t lds = gen_wgrad_out_lds(Si, Ci, ldc, Co, k)) {        // four output channels: GEMM form, the taps on the gradient
            int n_cu = 0;
            if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
            const long long R = (long long)N * Si;
            const int nsl = (int)std::max<long long>(1, std::min<long long>(std::min(GEN_WGRAD_OUT_SLICES_MAX, 2 * n_cu), R));
            const int np4 = (npair_out(Ci, k) + 3) / 4;
            if (hipError_t e = gen_wgrad_out_launch(st, in, dout, scratch, N, Si, Ci, ldc, k, nsl, np4, lds); e != hipSuccess) return e;
            hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, scratch, nsl, Ci, Ci_dst, Co,
                               k * k, alpha, gw, gb);
            return hipGetLastError();
        }
        if (const size_t lds = gen_wgrad_rows_lds(Si, Ci, ldc, Co, k)) {       // operands staged once per kernel row (round 5)
            // as many row slices as fill the chip with two blocks per CU (k x nsl x ngrp blocks; 64 slices left a third of the slots empty)
            int n_cu = 0;
            if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
            const int nsl = std::max(1, std::min(GEN_WGRAD_SLICES_MAX, 2 * n_cu / (k * ngrp)));
            const dim3 grid_r((unsigned)(k * nsl * ngrp));
            static std::atomic<unsigned> d3{0}, d5{0}, d7{0};
            if (k == 3) {
                if (hipError_t e = iod_set_max_lds((const void*)gen_wgrad_rows_kernel<3>, 80 * 1024, d3); e != hipSuccess) return e;
                hipLaunchKernelGGL((gen_wgrad_rows_kernel<3>), grid_r, dim3(256), lds, st, in, dout, scratch, N, Si, Ci, ldc, Co, nsl);
            } else if (k == 5) {
                if (hipError_t e = iod_set_max_lds((const void*)gen_wgrad_rows_kernel<5>, 80 * 1024, d5); e != hipSuccess) return e;
                hipLaunchKernelGGL((gen_wgrad_rows_kernel<5>), grid_r, dim3(256), lds, st, in, dout, scratch, N, Si, Ci, ldc, Co, nsl);
#ifdef IODINE_TILE_PROF
                if (getenv("IODINE_GEN_PROF")) {
                    const int nbk = std::min((int)grid_r.x, TP_MAXBLK);
                    std::vector<unsigned> hp((size_t)nbk * 8);
                    (void)hipStreamSynchronize(st);
                    (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_genw_prof), hp.size() * sizeof(unsigned));
                    static const char* names[8] = {"prologue", "commit", "barrier 1", "fetch issue", "MFMA blocks", "barrier 2", "-", "-"};
                    double sum[8] = {0}, tot = 0;
                    for (int b2 = 0; b2 < nbk; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[(size_t)b2 * 8 + i];
                    for (int i = 0; i < 8; ++i) tot += sum[i] / nbk;
                    fprintf(stderr, "[gen wgrad rows prof] ticks per block (%d slices), total %.0f:", nsl, tot);
                    for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.0f |", names[i], sum[i] / nbk);
                    fprintf(stderr, "\n");
                }
#endif
            } else {
                if (hipError_t e = iod_set_max_lds((const void*)gen_wgrad_rows_kernel<7>, 80 * 1024, d7); e != hipSuccess) return e;
                hipLaunchKernelGGL((gen_wgrad_rows_kernel<7>), grid_r, dim3(256), lds, st, in, dout, scratch, N, Si, Ci, ldc, Co, nsl);
            }
            hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, scratch, nsl, Ci, Ci_dst, Co,
                               k * k, alpha, gw, gb);
            return hipGetLastError();
        }
        const dim3 grid((unsigned)(k * k * GEN_WGRAD_SLICES * ngrp));
        if (k == 3) hipLaunchKernelGGL((gen_wgrad_mfma_kernel<3>), grid, dim3(256), 0, st, in, dout, scratch, N, Si, Ci, ldc, Co, GEN_WGRAD_SLICES);
        else if (k == 5) hipLaunchKernelGGL((gen_wgrad_mfma_kernel<5>), grid, dim3(256), 0, st, in, dout, scratch, N, Si, Ci, ldc, Co, GEN_WGRAD_SLICES);
        else hipLaunchKernelGGL((gen_wgrad_mfma_kernel<7>), grid, dim3(256), 0, st, in, dout, scratch, N, Si, Ci, ldc, Co, GEN_WGRAD_SLICES);
        hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, scratch, GEN_WGRAD_SLICES, Ci, Ci_dst, Co,
                           k * k, alpha, gw, gb);
        return hipGetLastError();
    }
    if (s == 2 && gen_s2_wgrad_ok(k, Ci, ldc, Co)) {
        int nsl = 0;
        if (hipError_t e = launch_gen_s2_wgrad(st, in, dout, scratch, N, Si, Ci, ldc, Co, k, GEN_WGRAD_SLICES_MAX, &nsl, chmask); e != hipSuccess) return e;
        hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, scratch, nsl, Ci, Ci_dst, Co,
                           k * k, alpha, gw, gb);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gen_conv_wgrad_partial_kernel, dim3(gen_blocks(per * GEN_WGRAD_SLICES)), dim3(256), 0, st, in, dout, scratch, N, Si,
                       So, Ci, ldc, Co, k, s, GEN_WGRAD_SLICES);
    hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, scratch, GEN_WGRAD_SLICES, Ci, Ci_dst, Co,
                       k * k, alpha, gw, gb);
    return hipGetLastError();
}

hipError_t launch_gen_identity(hipStream_t st, float* m, int rows, int L)
{
    hipLaunchKernelGGL(gen_identity_kernel, dim3(gen_blocks((size_t)rows * L)), dim3(256), 0, st, m, rows, L);
    return hipGetLastError();
}
