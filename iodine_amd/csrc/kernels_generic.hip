// Generic (any odd kernel size, any channel count) fp32 convolution kernels: the FALLBACK path of the library for ARCH
// combinations the tuned gfx950 kernels are not built for - KERNEL_SIZE 5 / 7 (the reference's default DEC.KERNEL_SIZE is 5,
// lib/config/defaults.py:100; configs/test.yaml:40,44) and CONV_CHAN other than 32 / 64.  Reference call sites: nn.Conv2d + F.elu of
// MultiLayerConv (iodine.py:570-594), the output conv (iodine.py:422,435), SpatialBroadcast (iodine.py:505-540) and their autograd.
//
// Two tiers.  Stride-1 convs (the decoder: ~96 % of the FLOPs) run on exact-fp32 MFMA kernels since round 4 (further down: LDS-resident
// weight slice per 16 output channels, 16 x 16 pixel tiles, v_mfma_f32_16x16x4_f32; weight gradient on v_mfma_f32_32x32x2_f32) whenever
// the weight slice fits LDS - 5 x 5 x 64 channels does, 7 x 7 up to 32 channels.  Everything else (stride 2 = the refinement stack,
// larger slices) runs on the scalar kernels right below: one thread per output element, plain fp32 FMAs in a fixed order, written for
// correctness.  All of it is deterministic.  The spatial-broadcast layer (decoder layer 0) has its own kernels (kernels_genl0.hip: the broadcast
// tensor is never built; until round 5 it was materialised as [N][P][L+2] and convolved like any other layer).  Every shipped / benchmarked configuration stays on the
// tuned path (iodine_api.cpp: `generic` is false for KERNEL_SIZE 3 with 32 / 64 channels).  Measured (MI355X, CLEVR shapes with
// DEC.KERNEL_SIZE 5, batch 4): training step 4977 -> 206 ms, reconstruct 2317 -> 96 ms against the scalar tier.
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

// gw[co][ci_dst][tap] += alpha * sum_slice partial (OIHW, ci < Ci_dst kept), gb[co] += alpha * sum_slice partial bias
__global__ void gen_conv_wgrad_reduce_kernel(const float* __restrict__ part, int nslice, int Ci, int Ci_dst, int Co, int kk, float alpha,
                                             float* __restrict__ gw, float* __restrict__ gb)
{
    const int per = kk * Ci * Co + Co;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= per) return;
    float acc = 0.f;
    for (int sl = 0; sl < nslice; ++sl) acc += part[(size_t)sl * per + e];
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
// Forward / data gradient (one kernel): implicit GEMM, M = 16 pixels of a tile row, N = 16 output channels, K = 4 channels of one tap.
// A persistent block owns ONE group of 16 output channels and keeps its whole weight slice [tap][reduction channel][16] in LDS
// (KS^2 x C x 64 B: 102 KB for 5 x 5 x 64) for all the tiles it processes - the weights are what a 5 x 5 conv re-reads most; the input
// halo of a 16 x 16 tile is staged per 4-channel chunk (double-buffered, one barrier per chunk = per 100 MFMAs of a wave).
//   W(tap, k, n) = wt[tapidx * sT + k * sK + n * sN]:  forward tapidx = tap, (sT, sK, sN) = (Ci Co, Co, 1) on the pack [tap][ci][co];
//   data gradient: the correlation with the flipped kernel, tapidx = KS^2 - 1 - tap, reduction over co, (sT, sK, sN) = (ldi Co, 1, Co).
// =====================================================================================================================================
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
    float* s_in = smem_g + (size_t)KK * Ckp * 16;             // [2][CCH][NPXP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // block -> (channel group, tile sequence).  Blocks b, b + 8, b + 16 ... share an XCD: the ncg blocks that need the same input tile at
    // the same time are put on ONE XCD (one L2 fetches the tile) whenever the counts divide; otherwise consecutive blocks take the groups.
    int cg, pb;
    const int nb = gridDim.x / ncg;
    if (nb % 8 == 0) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; cg = j % ncg; pb = (j / ncg) * 8 + x; }
    else { cg = blockIdx.x % ncg; pb = blockIdx.x / ncg; }
    if (pb >= nb) return;                                     // (grid = ncg * nb exactly; defensive)
    // ---- this block's weight slice -> LDS, once ----
    for (int e = tid; e < KK * Ckp * 16; e += 256) {
        const int n = e & 15, k = (e >> 4) % Ckp, tap = (e >> 4) / Ckp;
        const int co = cg * 16 + n;
        float v = 0.f;
        if (k < Ck && co < Cn) v = wt[(size_t)(flip ? KK - 1 - tap : tap) * sT + (size_t)k * sK + (size_t)co * sN];
        s_w[e] = v;
    }
    const int m = lane & 15, kq = lane >> 4;
    const int co0 = cg * 16 + 4 * kq;                         // the four output channels of this lane (D rows 4 kq + v, column = pixel m)
    float bv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = (bias && co0 + v < Cn) ? bias[co0 + v] : 0.f;
    const bool vec_out = (ldout & 3) == 0 && co0 + 3 < Cn;
    float4 rin[NLD];
    for (int t = pb; t < ntiles; t += nb) {
        const int tx = t % tiles, ty = (t / tiles) % tiles, n = t / (tiles * tiles);
        const float* in_n = in + (size_t)n * S * S * ldin;
        auto fetch = [&](int c) {
            int tv = tid;
            asm volatile("" : "+v"(tv));                       // (element -> (pixel, quad) arithmetic recomputed per chunk, not kept in registers)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tv + 256 * i;
                const int px = e / NQ, q = e % NQ;
                const int gy = ty * 16 - PAD + px / TW, gx = tx * 16 - PAD + px % TW, ch = c * CCH + 4 * q;
                const bool ok = e < NPX * NQ && (unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S && ch < Ck;
                rin[i] = ok ? *reinterpret_cast<const float4*>(in_n + ((size_t)gy * S + gx) * ldin + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto commit = [&](int buf) {
            int tv = tid;
            asm volatile("" : "+v"(tv));
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tv + 256 * i;
                if (e < NPX * NQ) {
                    float* d = s_in + (size_t)(buf * CCH + 4 * (e % NQ)) * NPXP + e / NQ;
                    d[0] = rin[i].x; d[NPXP] = rin[i].y; d[2 * NPXP] = rin[i].z; d[3 * NPXP] = rin[i].w;
                }
            }
        };
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                      // weights staged / previous tile's last chunk consumed
        fetch(0);
        commit(0);
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk) fetch(c + 1);                 // in flight under this chunk's MFMAs
            const float* si = s_in + (size_t)(c & 1) * CCH * NPXP + (size_t)kq * NPXP + (4 * wv) * TW + m;
            const float* sw = s_w + (size_t)(c * CCH + kq) * 16 + m;
            // All operands of a k-step (4 channels) are register-resident before its 4 KK MFMAs issue back to back: the KK weight values of
            // this lane and the (4 + KS - 1) x KS input values its four tile rows share between their taps (40 + 25 LDS reads at KS = 5).
            // Two operand sets: the reads of k-step h + 1 are issued before the MFMAs of k-step h, so that only the first k-step after a
            // barrier waits for the LDS.  One wave per SIMD (the weight slice fills the LDS) cannot hide a read-then-multiply chain behind
            // another wave.
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
            // MFMAs; the group barriers pin the pattern "two MFMAs of this k-step, one LDS read of the next")
            auto interleave = [&]() {
#pragma unroll
                for (int i = 0; i < 2 * KK; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            float bwA[KK], avA[RR][KS], bwB[KK], avB[RR][KS];
            ld(0, bwA, avA);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hh = 0; hh < NH; hh += 2) {
                if (hh + 1 < NH) ld(hh + 1, bwB, avB);
                mm(bwA, avA);
                interleave();
                if (hh + 1 < NH) {
                    if (hh + 2 < NH) ld(hh + 2, bwA, avA);
                    mm(bwB, avB);
                    interleave();
                }
            }
            if (c + 1 < nchunk) commit((c + 1) & 1);
            __syncthreads();
        }
        // D[channel 4 kq + v][pixel column m] of tile row 4 wv + r: one 16-byte store per lane and row
        const int x = tx * 16 + m;
        if (x < S && co0 < Cn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = ty * 16 + 4 * wv + r;
                if (y >= S) continue;
                const size_t o = (((size_t)n * S + y) * S + x) * ldout + co0;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = acc[r][j] + bv[j]; if (elu) v[j] = gen_elu(v[j]); }
                if (vec_out) {
                    if (aux) {
                        const float4 a = *reinterpret_cast<const float4*>(aux + o);
                        v[0] *= a.x > 0.f ? 1.f : a.x + 1.f; v[1] *= a.y > 0.f ? 1.f : a.y + 1.f;
                        v[2] *= a.z > 0.f ? 1.f : a.z + 1.f; v[3] *= a.w > 0.f ? 1.f : a.w + 1.f;
                    }
                    *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (co0 + j >= Cn) continue;
                        if (aux) { const float a = aux[o + j]; v[j] *= a > 0.f ? 1.f : a + 1.f; }
                        out[o + j] = v[j];
                    }
                }
            }
        }
    }
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
    constexpr int MAXV = 16;                                   // float4 per thread and staged row pair (launcher: fits)
    extern __shared__ __attribute__((aligned(16))) float smem_gw[];
    const int ky = blockIdx.x % KS, slice = (blockIdx.x / KS) % nslice, pgrp = blockIdx.x / (KS * nslice);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int nit = (Ci + 31) / 32, nct = (Co + 31) / 32, npair = nit * nct;
    const int Wp = S + 2 * PAD + 1, Sd = S + 1;               // staged pixels per row (+1: the pixel pair of an odd S reads a zero)
    float* s_a = smem_gw;                                      // [nit][Wp][32]
    float* s_d = smem_gw + (size_t)nit * Wp * 32;              // [nct][Sd][32]
    const int pair = pgrp * 4 + wv;
    const bool wave_on = pair < npair;
    const int cit = wave_on ? pair / nct : 0, cot = wave_on ? pair % nct : 0;
    const int co = cot * 32 + li;
    const bool cov = co < Co;
    const int per = KK * Ci * Co + Co;
    const long long R = (long long)N * S;
    const long long r0 = R * slice / nslice, r1 = R * (slice + 1) / nslice;
    // zero the planes once: channel pads, pixel margins and the extra pixel stay zero (staging only writes real elements)
    for (int e = tid; e < (nit * Wp + nct * Sd) * 32; e += 256) smem_gw[e] = 0.f;

    const int A4 = (Ci + 3) / 4, D4 = Co / 4;                 // float4 per pixel (Ci rounded up: the row stride covers the pad channels)
    const int na = S * A4, nd = S * D4;                       // float4 of a staged row (real pixels only)
    float4 rv[MAXV];
    auto fetch = [&](long long row) {
        const int y = (int)(row % S);
        const long long n = row / S;
        const int iy = y + ky - PAD;
        const bool rowv = (unsigned)iy < (unsigned)S;
        const float* ip = in + ((size_t)n * S + (rowv ? iy : 0)) * S * (size_t)ldc;
        const float* dp = dout + ((size_t)n * S + y) * S * (size_t)Co;
        int tv = tid;
        asm volatile("" : "+v"(tv));                           // (the element -> (pixel, quad) arithmetic is recomputed per row: hoisted it spills)
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int e = tv + 256 * k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < na) { if (rowv) v = *reinterpret_cast<const float4*>(ip + (size_t)(e / A4) * ldc + (e % A4) * 4); }
            else if (e < na + nd) { const int f = e - na; v = *reinterpret_cast<const float4*>(dp + (size_t)(f / D4) * Co + (f % D4) * 4); }
            rv[k] = v;
        }
    };
    auto commit = [&]() {
        int tv = tid;
        asm volatile("" : "+v"(tv));
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int e = tv + 256 * k;
            if (e < na) { const int x = e / A4, c4 = e % A4; *reinterpret_cast<float4*>(s_a + ((size_t)(c4 >> 3) * Wp + x + PAD) * 32 + (c4 & 7) * 4) = rv[k]; }
            else if (e < na + nd) { const int f = e - na, x = f / D4, c4 = f % D4; *reinterpret_cast<float4*>(s_d + ((size_t)(c4 >> 3) * Sd + x) * 32 + (c4 & 7) * 4) = rv[k]; }
        }
    };
    f32x16 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    float bsum = 0.f;
    const bool do_bias = ky == 0 && cit == 0 && wave_on;
    if (r0 < r1) fetch(r0);
    __syncthreads();                                           // planes zeroed
    for (long long row = r0; row < r1; ++row) {
        commit();
        __syncthreads();
        if (row + 1 < r1) fetch(row + 1);                      // in flight under this row's MFMAs
        const int iy = (int)(row % S) + ky - PAD;
        if (wave_on && ((unsigned)iy < (unsigned)S || do_bias)) {
            const float* ap = s_a + ((size_t)cit * Wp + kh) * 32 + li;
            const float* dp = s_d + ((size_t)cot * Sd + kh) * 32 + li;
#pragma unroll 4
            for (int x0 = 0; x0 < S; x0 += 2) {
                const float b = dp[(size_t)x0 * 32];
                float a[KS];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) a[kx] = ap[(size_t)(x0 + kx) * 32];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kx], b, acc[kx], 0, 0, 0);
                bsum += b;
            }
        }
        __syncthreads();                                       // every wave is done with the planes
    }
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

// LDS bytes of gen_wgrad_rows_kernel, 0 = the shape does not qualify (float4 staging, at most 16 float4 per thread and row, <= 80 KB: two blocks per CU)
inline size_t gen_wgrad_rows_lds(int S, int Ci, int ldc, int Co, int k)
{
    if (Co % 4 != 0 || ldc % 4 != 0 || ((Ci + 3) & ~3) > ldc) return 0;
    if ((size_t)S * ((Ci + 3) / 4 + Co / 4) > (size_t)16 * 256) return 0;
    const size_t b = ((size_t)((Ci + 31) / 32) * (S + 2 * (k / 2) + 1) + (size_t)((Co + 31) / 32) * (S + 1)) * 32 * sizeof(float);
    return b <= 80 * 1024 ? b : 0;
}

inline unsigned gen_blocks(size_t total) { return (unsigned)((total + 255) / 256); }

// LDS bytes of the MFMA form with chunks of cch channels: weight slice + two channel-major halo chunks; 0 = does not apply (stride 2, or the
// slice does not fit)
inline size_t gen_mfma_lds_cch(int k, int Ck, int s, int cch)
{
    if (s != 1 || (k != 3 && k != 5 && k != 7)) return 0;
    const int Ckp = (Ck + cch - 1) / cch * cch, TW = 16 + k - 1, NPXP = (TW * TW + 15) / 32 * 32 + 16;
    const size_t b = ((size_t)k * k * Ckp * 16 + (size_t)2 * NPXP * cch) * sizeof(float);
    return b <= 160 * 1024 ? b : 0;
}
// chunk width: the smallest of 4 / 8 / 16 that covers the reduction channels, stepping down while the LDS does not fit; 0 = the scalar kernels
inline int gen_mfma_cch(int k, int Ck, int s)
{
    int cch = Ck <= 4 ? 4 : Ck <= 8 ? 8 : 16;
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
    return hipGetLastError();
}

template <int KS>
hipError_t gen_mfma_launch(hipStream_t st, const float* in, const float* wt, const float* bias, const float* aux, float* out, int N, int S,
                           int Ck, int ldin, int Cn, int ldout, int flip, int sT, int sK, int sN, int elu, size_t lds)
{
    switch (gen_mfma_cch(KS, Ck, 1)) {
    case 16: return gen_mfma_launch_cch<KS, 16>(st, in, wt, bias, aux, out, N, S, Ck, ldin, Cn, ldout, flip, sT, sK, sN, elu, lds);
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
                               int ldc, int Co, int k, int s, int elu)
{
    if (const size_t lds = gen_mfma_lds(k, Ci, ldc, s)) {
        // [tap][ci][co] pack: W(tap, k = ci, n = co)
        if (k == 3) return gen_mfma_launch<3>(st, in, wt, bias, nullptr, out, N, Si, Ci, ldc, Co, Co, 0, Ci * Co, Co, 1, elu, lds);
        if (k == 5) return gen_mfma_launch<5>(st, in, wt, bias, nullptr, out, N, Si, Ci, ldc, Co, Co, 0, Ci * Co, Co, 1, elu, lds);
        return gen_mfma_launch<7>(st, in, wt, bias, nullptr, out, N, Si, Ci, ldc, Co, Co, 0, Ci * Co, Co, 1, elu, lds);
    }
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
    const int So = (Si - 1) / s + 1;
    const size_t total = (size_t)N * Si * Si * Ci;
    hipLaunchKernelGGL(gen_conv_dgrad_kernel, dim3(gen_blocks(total)), dim3(256), 0, st, dout, wt, aux, din, Si, So, Ci, ldi, Co, k, s, total);
    return hipGetLastError();
}

size_t gen_wgrad_scratch_floats(int Ci, int Co, int k) { return (size_t)GEN_WGRAD_SLICES_MAX * ((size_t)k * k * Ci * Co + Co); }

hipError_t launch_gen_conv_wgrad(hipStream_t st, const float* in, const float* dout, float* scratch, int N, int Si, int Ci, int ldc,
                                 int Ci_dst, int Co, int k, int s, float alpha, float* gw, float* gb)
{
    const int So = (Si - 1) / s + 1;
    const size_t per = (size_t)k * k * Ci * Co + Co;
    if (s == 1 && (k == 3 || k == 5 || k == 7)) {
        const int npair = ((Ci + 31) / 32) * ((Co + 31) / 32), ngrp = (npair + 3) / 4;
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
            } else {
                if (hipError_t e = iod_set_max_lds((const void*)gen_wgrad_rows_kernel<7>, 80 * 1024, d7); e != hipSuccess) return e;
                hipLaunchKernelGGL((gen_wgrad_rows_kernel<7>), grid_r, dim3(256), lds, st, in, dout, scratch, N, Si, Ci, ldc, Co, nsl);
            }
            hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3(gen_blocks(per)), dim3(256), 0, st, scratch, nsl, Ci, Ci_dst, Co,
                               k * k, alpha, gw, gb);
            return hipGetLastError();
        }
        const dim3 grid((unsigned)(k * k * GEN_WGRAD_SLICES * ngrp));
        if (k == 3) hipLaunchKernelGGL((gen_wgrad_mfma_kernel<3>), grid, dim3(256), 0, st, in, dout, scratch, N, Si, Ci, ldc, Co, GEN_WGRAD_SLICES);
        else if (k == 5) hipLaunchKernelGGL((gen_wgrad_mfma_kernel<5>), grid, dim3(256), 0, st, in, dout, scratch, N, Si, Ci, ldc, Co, GEN_WGRAD_SLICES);
        else hipLaunchKernelGGL((gen_wgrad_mfma_kernel<7>), grid, dim3(256), 0, st, in, dout, scratch, N, Si, Ci, ldc, Co, GEN_WGRAD_SLICES);
        hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3(gen_blocks(per)), dim3(256), 0, st, scratch, GEN_WGRAD_SLICES, Ci, Ci_dst, Co,
                           k * k, alpha, gw, gb);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gen_conv_wgrad_partial_kernel, dim3(gen_blocks(per * GEN_WGRAD_SLICES)), dim3(256), 0, st, in, dout, scratch, N, Si,
                       So, Ci, ldc, Co, k, s, GEN_WGRAD_SLICES);
    hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3(gen_blocks(per)), dim3(256), 0, st, scratch, GEN_WGRAD_SLICES, Ci, Ci_dst, Co,
                       k * k, alpha, gw, gb);
    return hipGetLastError();
}

hipError_t launch_gen_identity(hipStream_t st, float* m, int rows, int L)
{
    hipLaunchKernelGGL(gen_identity_kernel, dim3(gen_blocks((size_t)rows * L)), dim3(256), 0, st, m, rows, L);
    return hipGetLastError();
}
