// Per-pixel mixture kernels: mask softmax, Gaussian log-likelihood, log-mixture, the closed-form
// inner gradients and the 17-channel refinement input with its layer-norm statistics.
//
// Reference call sites (lib/modeling/iodine.py):
//   elbo():               softmax :185, gaussian_log_likelihood :210,661-666, logsumexp :213-216, .mean(0).sum() :220
//   (B*elbo).backward():  mean.grad / mask.grad :90,137,181,187  -> closed forms
//   get_input_encoding(): channel order :277-340, 5-D layernorm :385-394
// One thread owns one pixel and keeps all K slots of it in registers (the K-softmax, the
// log-mixture and the leave-one-out term couple the slots at every pixel).  HBM-bound.
#include "common.h"

#define PIX_BLOCK 256

#include "pixel_terms.h"

// sum of a double over the wave (valid in every lane).  Lane swaps / DPP in the VALU on the two 32-bit halves: __shfl_down on a
// double is two ds_bpermute per step, and the pass-1 kernel reduces 6K + 3 statistics per wave for only two pixels per thread.
IOD_DEVINL double wave_sum_d(double v)
{
    auto halves = [](double x, unsigned& lo, unsigned& hi) { const unsigned long long u = __double_as_longlong(x); lo = (unsigned)u; hi = (unsigned)(u >> 32); };
    auto join = [](unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
    unsigned lo, hi;
    {
        halves(v, lo, hi);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = join(a[0], b[0]) + join(a[1], b[1]);
    }
    {
        halves(v, lo, hi);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = join(a[0], b[0]) + join(a[1], b[1]);
    }
#define IOD_ROW_ROR_ADD(ctrl)                                                                                  \
    {                                                                                                          \
        halves(v, lo, hi);                                                                                     \
        v += join(__builtin_amdgcn_update_dpp(0u, lo, ctrl, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0u, hi, ctrl, 0xf, 0xf, false)); \
    }
    IOD_ROW_ROR_ADD(0x128) IOD_ROW_ROR_ADD(0x124) IOD_ROW_ROR_ADD(0x122) IOD_ROW_ROR_ADD(0x121)       // row_ror:8, 4, 2, 1
#undef IOD_ROW_ROR_ADD
    return v;
}

// ---- pass 1: ELBO log-likelihood partials, gradient wrt decoder output, layer-norm partial sums ----
// part layout per (b, block): [0] ll, [1] like.s1, [2] like.s2, then per k: g1.s1, g1.s2, g2.s1, g2.s2, loo.s1, loo.s2
template <int K, bool STRICT>
__global__ __launch_bounds__(PIX_BLOCK)
void pixel_pass1_kernel(const float4* __restrict__ x4, const float4* __restrict__ dec, float4* __restrict__ g,
                        double* __restrict__ part, int P, int ppb, float inv2s2, float invs2, float lconst)
{
    constexpr int NST = 6 * K + 3;
    const int nblk = gridDim.x, b = blockIdx.y, blk = blockIdx.x;
    const int tid = threadIdx.x;
    const float4* dec_b = dec + (size_t)b * K * P;
    float4* g_b = g + (size_t)b * K * P;

    float st[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) st[i] = 0.f;

    const int pend = min(P, (blk + 1) * ppb);
    for (int p = blk * ppb + tid; p < pend; p += PIX_BLOCK) {
        PixelTerms<K> t;
        pixel_terms<K, STRICT>(x4[(size_t)b * P + p], dec_b, (size_t)P, (size_t)p, inv2s2, invs2, lconst, t);
        float tg = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) tg += t.m[k] * t.g2[k];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float4 o;
            o.x = t.g1[k][0] * t.mu[k][0] * (1.f - t.mu[k][0]);
            o.y = t.g1[k][1] * t.mu[k][1] * (1.f - t.mu[k][1]);
            o.z = t.g1[k][2] * t.mu[k][2] * (1.f - t.mu[k][2]);
            o.w = t.m[k] * (t.g2[k] - tg);
            g_b[(size_t)k * P + p] = o;
            const float loo = t.loo[k];
            st[3 + 6 * k + 0] += t.g1[k][0] + t.g1[k][1] + t.g1[k][2];
            st[3 + 6 * k + 1] += t.g1[k][0] * t.g1[k][0] + t.g1[k][1] * t.g1[k][1] + t.g1[k][2] * t.g1[k][2];
            st[3 + 6 * k + 2] += t.g2[k];
            st[3 + 6 * k + 3] += t.g2[k] * t.g2[k];
            st[3 + 6 * k + 4] += loo;
            st[3 + 6 * k + 5] += loo * loo;
        }
        st[0] += t.ll_sum;
        st[1] += t.like;
        st[2] += t.like * t.like;
    }

    // Block sum of the NST statistics in fp64.  Through LDS, not DPP: 6 K + 3 wave reductions in fp64 were ~1800 instructions per wave, more
    // than one pixel_terms; here thread (statistic i, wave w) adds the 64 per-thread fp32 partials of that wave (row stride 257 and a start
    // rotated by 16 w: conflict-free), then NST threads add the four wave sums.  A fixed order: deterministic.
    __shared__ float s_part[NST][PIX_BLOCK + 1];
    __shared__ double s_red[NST][PIX_BLOCK / 64];
#pragma unroll
    for (int i = 0; i < NST; ++i) s_part[i][tid] = st[i];
    __syncthreads();
    for (int q = tid; q < NST * (PIX_BLOCK / 64); q += PIX_BLOCK) {
        const int i = q / (PIX_BLOCK / 64), w = q % (PIX_BLOCK / 64);
        const float* row = s_part[i] + w * 64;
        double v = 0.0;
#pragma unroll 8
        for (int j = 0; j < 64; ++j) v += (double)row[(j + 16 * w) & 63];
        s_red[i][w] = v;
    }
    __syncthreads();
    if (tid < NST) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < PIX_BLOCK / 64; ++w) v += s_red[tid][w];
        part[((size_t)b * nblk + blk) * NST + tid] = v;
    }
}

// ---- finalize: fixed-order sum of block partials -> per-image LL, per-slot LN (mean, 1/(std+1e-5)) ----
// lnstat[n][8] = {g1.mean, g1.inv, g2.mean, g2.inv, loo.mean, loo.inv, like.mean, like.inv}
__global__ void pixel_finalize_kernel(const double* __restrict__ part, int nblk, int K, int P, int use_ln,
                                      float* __restrict__ lnstat, float* __restrict__ ll_img)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NST = 6 * K + 3;
    __shared__ double s_sum[6 * 16 + 3];
    if (tid < NST) {
        // fixed order, eight independent loads in flight (a plain running sum is a chain of nblk dependent HBM round trips: 11 us)
        double v = 0.0;
        int i = 0;
        for (; i + 7 < nblk; i += 8) {
            double q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = part[((size_t)b * nblk + i + j) * NST + tid];
#pragma unroll
            for (int j = 0; j < 8; ++j) v += q[j];
        }
        for (; i < nblk; ++i) v += part[((size_t)b * nblk + i) * NST + tid];
        s_sum[tid] = v;
    }
    __syncthreads();
    if (tid == 0) ll_img[b] = (float)s_sum[0];
    if (tid < K) {
        float* o = lnstat + ((size_t)b * K + tid) * 8;
        const double cnt[4] = {3.0 * P, (double)P, (double)P, (double)P};
        const double s1[4] = {s_sum[3 + 6 * tid + 0], s_sum[3 + 6 * tid + 2], s_sum[3 + 6 * tid + 4], s_sum[1]};
        const double s2[4] = {s_sum[3 + 6 * tid + 1], s_sum[3 + 6 * tid + 3], s_sum[3 + 6 * tid + 5], s_sum[2]};
        for (int j = 0; j < 4; ++j) {
            const double mean = s1[j] / cnt[j];
            double var = s2[j] / cnt[j] - mean * mean;
            if (var < 0.0) var = 0.0;
            const float sd = (float)sqrt(var);
            o[2 * j + 0] = use_ln ? (float)mean : 0.f;
            o[2 * j + 1] = use_ln ? 1.f / (sd + 1e-5f) : 1.f;
        }
    }
}

// ---- finalize + KL against N(0, 1) (iodine.py:653-659,191-193) + ELBO assembly in ONE launch (three until round 4: 5 us each behind a
// 50 us kernel, six times per step).  Block b: image b.  img_terms[b] = {ll_b, kl_b}; the LAST block to finish (device-scope counter, reset by
// that block) forms scal = {elbo, kl, ll} as the means over the images in fixed order - same arithmetic as kl_image_kernel /
// elbo_mean_kernel (kernels_misc.hip), which remain for the public iodine_elbo path.
__global__ __launch_bounds__(256)
void pixel_finalize_elbo_kernel(const double* __restrict__ part, int nblk, int K, int P, int use_ln, float* __restrict__ lnstat,
                                float* __restrict__ ll_img, const float* __restrict__ pm, const float* __restrict__ plv, int KL_,
                                float* __restrict__ img_terms, float* __restrict__ scal, unsigned* __restrict__ counter)
{
    const int b = blockIdx.x, tid = threadIdx.x, B = gridDim.x;
    const int NST = 6 * K + 3;
    __shared__ double s_sum[6 * 16 + 3];
    __shared__ float s_buf[4];
    __shared__ unsigned s_last;
    if (tid < NST) {
        double v = 0.0;
        int i = 0;
        for (; i + 7 < nblk; i += 8) {
            double q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = part[((size_t)b * nblk + i + j) * NST + tid];
#pragma unroll
            for (int j = 0; j < 8; ++j) v += q[j];
        }
        for (; i < nblk; ++i) v += part[((size_t)b * nblk + i) * NST + tid];
        s_sum[tid] = v;
    }
    float s = 0.f;
    for (int i = tid; i < KL_; i += 256) {
        const float mu = pm[(size_t)b * KL_ + i], lv = plv[(size_t)b * KL_ + i];
        s += 0.5f * (expf(lv) + mu * mu - 1.f - lv);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((tid & 63) == 0) s_buf[tid >> 6] = s;
    __syncthreads();
    if (tid < K) {
        float* o = lnstat + ((size_t)b * K + tid) * 8;
        const double cnt[4] = {3.0 * P, (double)P, (double)P, (double)P};
        const double s1[4] = {s_sum[3 + 6 * tid + 0], s_sum[3 + 6 * tid + 2], s_sum[3 + 6 * tid + 4], s_sum[1]};
        const double s2[4] = {s_sum[3 + 6 * tid + 1], s_sum[3 + 6 * tid + 3], s_sum[3 + 6 * tid + 5], s_sum[2]};
        for (int j = 0; j < 4; ++j) {
            const double mean = s1[j] / cnt[j];
            double var = s2[j] / cnt[j] - mean * mean;
            if (var < 0.0) var = 0.0;
            const float sd = (float)sqrt(var);
            o[2 * j + 0] = use_ln ? (float)mean : 0.f;
            o[2 * j + 1] = use_ln ? 1.f / (sd + 1e-5f) : 1.f;
        }
    }
    if (tid == 0) {
        const float ll = (float)s_sum[0];
        const float kl = ((s_buf[0] + s_buf[1]) + s_buf[2]) + s_buf[3];       // block_sum_f's order
        ll_img[b] = ll;
        img_terms[2 * b] = ll; img_terms[2 * b + 1] = kl;
        __threadfence();                                                      // release: the two terms before the ticket
        s_last = atomicAdd(counter, 1u) == (unsigned)(B - 1);
    }
    __syncthreads();
    if (s_last) {                                                             // block-uniform
        __threadfence();                                                      // acquire: the other blocks' terms
        __shared__ float s_t[2 * 256];
        double ll = 0.0, kl = 0.0;
        for (int i0 = 0; i0 < B; i0 += 256) {                                 // loads in parallel, sums in image order (elbo_mean_kernel's)
            if (i0 + tid < B) {
                s_t[2 * tid] = __builtin_nontemporal_load(img_terms + 2 * (i0 + tid));
                s_t[2 * tid + 1] = __builtin_nontemporal_load(img_terms + 2 * (i0 + tid) + 1);
            }
            __syncthreads();
            if (tid == 0)
                for (int i = 0; i < min(256, B - i0); ++i) { ll += s_t[2 * i]; kl += s_t[2 * i + 1]; }
            __syncthreads();
        }
        if (tid == 0) {
            ll /= B; kl /= B;
            scal[0] = (float)(ll - kl); scal[1] = (float)kl; scal[2] = (float)ll;
            *counter = 0u;                                                    // (next launch: stream order)
        }
    }
}

// ---- pass 2: write the refinement input, NHWC with 20 channels (17 + 3 zero pad) -----------------
// SPLIT form (split first refinement layer): the 11 channels that differ between the slots of an image go to
//   enc[n][p][12]    = mean rgb, mask, mask logit, mask posterior, LN(d mean) rgb, LN(d mask), LN(leave-one-out), 0
// and the 6 channels every slot of an image shares, once per image, to
//   enc_sh[b][p][8]  = image rgb, LN(pixel likelihood), coordinate x, coordinate y, 0, 0
// (294 -> 176 + 17 MB per iteration at cfg3; the shared part is convolved once per image instead of once per slot).
template <int K, bool SPLIT, bool STRICT>
__global__ __launch_bounds__(PIX_BLOCK)
void pixel_pass2_kernel(const float4* __restrict__ x4, const float4* __restrict__ dec,
                        const float* __restrict__ lnstat, const float* __restrict__ lin, float* __restrict__ enc,
                        float* __restrict__ enc_sh, int P, int S, int ppb, float inv2s2, float invs2, float lconst, unsigned chmask)
{
    // ARCH.ENCODING subsets: bit c of chmask = internal channel c (code order of iodine.py:277-340) is part of the encoding.  An absent
    // channel is written as 0 - its weights are zero too (enc_expand_weights), but 0 * Inf would still be NaN, and the reference never
    // computes such a channel (e.g. the leave-one-out likelihood, a cancellation divided by 1e-5).
    auto on = [&](int c, float v) { return (chmask >> c) & 1u ? v : 0.f; };
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const float4* dec_b = dec + (size_t)b * K * P;
    __shared__ float s_ln[K * 8];
    for (int i = tid; i < K * 8; i += PIX_BLOCK) s_ln[i] = lnstat[(size_t)b * K * 8 + i];
    __syncthreads();
    // The 20-channel rows are transposed through a wave-private LDS tile so that the wave writes its 64 pixels of a slot
    // as 5 KB of CONTIGUOUS float4s (the direct form stored 16 bytes per lane at an 80-byte stride, five passes over the
    // same 40 cache lines per slot).
    __shared__ __attribute__((aligned(16))) float s_tr[PIX_BLOCK / 64][64 * 20];
    const int lane = tid & 63, wv = tid >> 6;
    float4* tw = reinterpret_cast<float4*>(s_tr[wv] + lane * 20);
    const float4* trd = reinterpret_cast<const float4*>(s_tr[wv]);
    const int pend = min(P, (blk + 1) * ppb);
    for (int p0 = blk * ppb + wv * 64; p0 < pend; p0 += PIX_BLOCK) {       // wave-uniform
        const int nvalid = min(64, pend - p0);
        const int p = p0 + min(lane, nvalid - 1);                          // idle lanes recompute the last pixel
        PixelTerms<K> t;
        const float4 xv = x4[(size_t)b * P + p];
        pixel_terms<K, STRICT>(xv, dec_b, (size_t)P, (size_t)p, inv2s2, invs2, lconst, t);
        float psum = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) psum += t.pk[k];
        const float cx = lin[p % S], cy = lin[p / S];
        if constexpr (SPLIT) {
            float4* tw8 = reinterpret_cast<float4*>(s_tr[wv] + lane * 8);
            tw8[0] = make_float4(on(0, xv.x), on(1, xv.y), on(2, xv.z), on(13, (t.like - s_ln[6]) * s_ln[7]));   // LN statistics of slot 0: the same for every slot
            tw8[1] = make_float4(on(15, cx), on(16, cy), 0.f, 0.f);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float4* o = reinterpret_cast<float4*>(enc_sh + ((size_t)b * P + p0) * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = lane + 64 * j;
                if (q < nvalid * 2) o[q] = trd[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float* ln = s_ln + k * 8;
                const float loo = t.loo[k];
                float4* tw12 = reinterpret_cast<float4*>(s_tr[wv] + lane * 12);
                tw12[0] = make_float4(on(3, t.mu[k][0]), on(4, t.mu[k][1]), on(5, t.mu[k][2]), on(6, t.m[k]));
                tw12[1] = make_float4(on(7, t.logit[k]), on(8, t.pk[k] / psum), on(9, (t.g1[k][0] - ln[0]) * ln[1]), on(10, (t.g1[k][1] - ln[0]) * ln[1]));
                tw12[2] = make_float4(on(11, (t.g1[k][2] - ln[0]) * ln[1]), on(12, (t.g2[k] - ln[2]) * ln[3]), on(14, (loo - ln[4]) * ln[5]), 0.f);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float4* ok = reinterpret_cast<float4*>(enc + (((size_t)b * K + k) * P + p0) * 12);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int q = lane + 64 * j;
                    if (q < nvalid * 3) ok[q] = trd[q];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float* ln = s_ln + k * 8;
            const float loo = t.loo[k];
            tw[0] = make_float4(on(0, xv.x), on(1, xv.y), on(2, xv.z), on(3, t.mu[k][0]));
            tw[1] = make_float4(on(4, t.mu[k][1]), on(5, t.mu[k][2]), on(6, t.m[k]), on(7, t.logit[k]));
            tw[2] = make_float4(on(8, t.pk[k] / psum), on(9, (t.g1[k][0] - ln[0]) * ln[1]), on(10, (t.g1[k][1] - ln[0]) * ln[1]),
                                on(11, (t.g1[k][2] - ln[0]) * ln[1]));
            tw[3] = make_float4(on(12, (t.g2[k] - ln[2]) * ln[3]), on(13, (t.like - ln[6]) * ln[7]), on(14, (loo - ln[4]) * ln[5]), on(15, cx));
            tw[4] = make_float4(on(16, cy), 0.f, 0.f, 0.f);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float4* o = reinterpret_cast<float4*>(enc + (((size_t)b * K + k) * P + p0) * 20);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int q = lane + 64 * j;
                if (q < nvalid * 5) o[q] = trd[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// ---- final decode outputs (IODINE.decode, lib/modeling/iodine.py:59-71): NCHW pred / mask / mean ----
template <int K>
__global__ __launch_bounds__(PIX_BLOCK)
void final_out_kernel(const float4* __restrict__ dec, float* __restrict__ pred, float* __restrict__ mask,
                      float* __restrict__ mean, float* __restrict__ logits, int P)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * PIX_BLOCK + threadIdx.x;
    if (p >= P) return;
    const float4* dec_b = dec + (size_t)b * K * P;
    float4 d[K];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { d[k] = dec_b[(size_t)k * P + p]; mx = fmaxf(mx, d[k].w); }
    float m[K], den = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { m[k] = expf(d[k].w - mx); den += m[k]; }
    const float rden = 1.f / den;
    float pr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        m[k] *= rden;
        const float mu[3] = {sigmoidf_(d[k].x), sigmoidf_(d[k].y), sigmoidf_(d[k].z)};
        const size_t n = (size_t)b * K + k;
        if (mask) mask[n * P + p] = m[k];
        if (logits) logits[n * P + p] = d[k].w;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (mean) mean[(n * 3 + c) * P + p] = mu[c];
            pr[c] += m[k] * mu[c];
        }
    }
    if (pred)
#pragma unroll
        for (int c = 0; c < 3; ++c) pred[((size_t)b * 3 + c) * P + p] = pr[c];
}

// ---- launchers --------------------------------------------------------------------------------
#define FOR_EACH_K(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)      // (round 6: 13 .. 16; 6 K + 3 <= 99 statistics)

int pixel_blocks_per_image(int P) { return (P + 2 * PIX_BLOCK - 1) / (2 * PIX_BLOCK); }   // 2 pixels / thread

hipError_t launch_pixel_pass1(hipStream_t st, const float* x4, const float* dec, float* g, double* part, int B,
                              int K, int P, float sigma, int strict)
{
    IOD_XSKIP(8);
    const int nblk = pixel_blocks_per_image(P), ppb = (P + nblk - 1) / nblk;
    const float inv2s2 = 1.f / (2.f * sigma * sigma), invs2 = 1.f / (sigma * sigma);
    const float lconst = (float)(-log((double)sigma) - 0.5 * log(2.0 * M_PI));
    switch (K) {
#define CASE(KK) case KK: \
        if (strict) hipLaunchKernelGGL((pixel_pass1_kernel<KK, true>), dim3(nblk, B), dim3(PIX_BLOCK), 0, st, \
            (const float4*)x4, (const float4*)dec, (float4*)g, part, P, ppb, inv2s2, invs2, lconst); \
        else hipLaunchKernelGGL((pixel_pass1_kernel<KK, false>), dim3(nblk, B), dim3(PIX_BLOCK), 0, st, \
            (const float4*)x4, (const float4*)dec, (float4*)g, part, P, ppb, inv2s2, invs2, lconst); \
        break;
        FOR_EACH_K(CASE)
#undef CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_pixel_finalize(hipStream_t st, const double* part, int B, int K, int P, int use_ln,
                                 float* lnstat, float* ll_img)
{
    IOD_XSKIP(8);
    hipLaunchKernelGGL(pixel_finalize_kernel, dim3(B), dim3(128), 0, st, part, pixel_blocks_per_image(P), K, P,
                       use_ln, lnstat, ll_img);
    return hipGetLastError();
}

hipError_t launch_pixel_finalize_elbo(hipStream_t st, const double* part, int B, int K, int P, int use_ln, float* lnstat, float* ll_img,
                                      const float* pm, const float* plv, int L, float* img_terms, float* scal, unsigned* counter)
{
    IOD_XSKIP(8);
    if (6 * K + 3 > 99 || !counter) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pixel_finalize_elbo_kernel, dim3(B), dim3(256), 0, st, part, pixel_blocks_per_image(P), K, P, use_ln, lnstat, ll_img,
                       pm, plv, K * L, img_terms, scal, counter);
    return hipGetLastError();
}

hipError_t launch_pixel_pass2(hipStream_t st, const float* x4, const float* dec, const float* lnstat,
                              const float* lin, float* enc, int B, int K, int S, float sigma, float* enc_sh, unsigned chmask, int strict)
{
    IOD_XSKIP(8);
    const int P = S * S;
    const int nblk = pixel_blocks_per_image(P), ppb = (P + nblk - 1) / nblk;
    const float inv2s2 = 1.f / (2.f * sigma * sigma), invs2 = 1.f / (sigma * sigma);
    const float lconst = (float)(-log((double)sigma) - 0.5 * log(2.0 * M_PI));
    switch (K) {
#define P2(KQ, SP, ST) hipLaunchKernelGGL((pixel_pass2_kernel<KQ, SP, ST>), dim3(nblk, B), dim3(PIX_BLOCK), 0, st, \
            (const float4*)x4, (const float4*)dec, lnstat, lin, enc, enc_sh, P, S, ppb, inv2s2, invs2, lconst, chmask)
#define CASE(KK) case KK: \
        if (enc_sh) { if (strict) P2(KK, true, true); else P2(KK, true, false); } \
        else { if (strict) P2(KK, false, true); else P2(KK, false, false); } \
        break;
        FOR_EACH_K(CASE)
#undef CASE
#undef P2
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_final_out(hipStream_t st, const float* dec, float* pred, float* mask, float* mean, float* logits,
                            int B, int K, int P)
{
    const dim3 grid((P + PIX_BLOCK - 1) / PIX_BLOCK, B);
    switch (K) {
#define CASE(KK) case KK: hipLaunchKernelGGL((final_out_kernel<KK>), grid, dim3(PIX_BLOCK), 0, st, \
        (const float4*)dec, pred, mask, mean, logits, P); break;
        FOR_EACH_K(CASE)
#undef CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
