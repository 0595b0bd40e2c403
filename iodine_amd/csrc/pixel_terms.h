// Per-pixel mixture terms shared by the pixel kernels (kernels_pixel.hip) and the fused first refinement layer (kernels_refl0.hip):
// ONE definition, so that every kernel that needs the 17-channel encoding computes bit-identical values (the function switches fp
// contraction off for exactly that reason - see the comments inside).  Reference: lib/modeling/iodine.py:185-216, 277-331.
#pragma once
#include "common.h"

template <int K>
struct PixelTerms {
    float mu[K][3];       // sigmoid(rgb)
    float m[K];           // mask = softmax_k(logit)
    float logit[K];
    float g1[K][3];       // d(B*ELBO)/d mean
    float g2[K];          // d(B*ELBO)/d mask
    float pk[K];          // exp(sum_c l_kc)  (un-stabilised, as the reference)
    float ll_sum;         // sum_c logsumexp_k(log(m_k + 1e-12) + l_kc)
    float like;           // exp(ll_sum)
    float mix;            // sum_k m_k * pk_k
    float loo[K];         // leave-one-out likelihood (iodine.py:321-328), see pixel_terms
};

// Round 5: of the ~2080 VALU instructions this function took per pixel at K = 7, 740 were the range / denormal handling around the 57
// v_exp_f32 of libm's expf and 530 the IEEE sequences of 53 divisions - and every kernel that inlines it (pass 1, pass 2, the fused first
// refinement layer) is VALU-issue-bound.  Where the argument is bounded and the result enters a sum next to a term of order 1 - the
// sigmoids, the max-subtracted softmax over the slots, the max-subtracted responsibilities - exp is now v_exp_f32(x log2 e) (relative
// error <= |x| 2^-24 + 1 ulp; results below the normal range flush to 0 next to a 1) and a / b is a * v_rcp_f32(b) (1 ulp + 1 rounding;
// 1 / inf = 0 and NaN propagate like the IEEE sequence).  The UN-stabilised terms of the reference keep libm's expf and the IEEE
// division: p_k = exp(sum_c l_kc) and exp(ll_sum) (arguments down to -140: denormal results matter for WHERE the 0 / 0 of
// mask_posterior appears, iodine.py:286-293) and the leave-one-out channel (the reference's exact sequence of rounded operations, below).
// Round 6: STRICT = the strict path (option conv_precision 0, "the reference's arithmetic"): libm expf and IEEE division
// everywhere, like ATen - the hardware approximations above are a property of the DEFAULT path only.
template <bool STRICT> IOD_DEVINL float pt_exp_bounded(float x) { if constexpr (STRICT) return expf(x); else return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
template <bool STRICT> IOD_DEVINL float pt_rcp(float d) { if constexpr (STRICT) return 1.f / d; else return __builtin_amdgcn_rcpf(d); }
template <bool STRICT> IOD_DEVINL float pt_sigmoid(float v) { return pt_rcp<STRICT>(1.f + pt_exp_bounded<STRICT>(-v)); }

// dv[k] = decoder output (rgb logits, mask logit) of slot k at this pixel
template <int K, bool STRICT = false>
IOD_DEVINL void pixel_terms_core(const float4 xv, const float4 (&dv)[K], float inv2s2, float invs2, float lconst, PixelTerms<K>& t)
{
    // No fp contraction in here: pass 1 (layer-norm statistics) and pass 2 (the values that get normalised) inline this function
    // separately, and hipcc's default -ffp-contract=fast fuses multiplies into adds differently per instantiation.  A 1-ulp
    // difference upstream is harmless everywhere except in the leave-one-out channel below, whose value at a saturated mask is
    // rounding noise x 1e5: a noise spike that pass 2 sees but pass 1's statistics do not contain is normalised to 100 sigma
    // instead of being absorbed by the standard deviation (as in the reference, which computes the channel once).
#pragma clang fp contract(off)
    const float xs[3] = {xv.x, xv.y, xv.z};
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4 d = dv[k];
        t.mu[k][0] = pt_sigmoid<STRICT>(d.x); t.mu[k][1] = pt_sigmoid<STRICT>(d.y); t.mu[k][2] = pt_sigmoid<STRICT>(d.z);
        t.logit[k] = d.w;
        mx = fmaxf(mx, d.w);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { t.m[k] = pt_exp_bounded<STRICT>(t.logit[k] - mx); den += t.m[k]; }
    const float rden = pt_rcp<STRICT>(den);
    float lm[K], rme[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        t.m[k] *= rden;
        const float me = t.m[k] + 1e-12f;
        lm[k] = logf(me);
        rme[k] = pt_rcp<STRICT>(me);                                   // 1 / (m_k + 1e-12): d log(m + 1e-12) / dm, once per slot
        t.g2[k] = 0.f;
    }
    float lsum[K];
#pragma unroll
    for (int k = 0; k < K; ++k) lsum[k] = 0.f;
    t.ll_sum = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a[K];
        float amax = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d = xs[c] - t.mu[k][c];
            const float l = -(d * d) * inv2s2 + lconst;
            lsum[k] += l;
            a[k] = lm[k] + l;
            amax = fmaxf(amax, a[k]);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { a[k] = pt_exp_bounded<STRICT>(a[k] - amax); s += a[k]; }
        t.ll_sum += amax + logf(s);
        const float rs = pt_rcp<STRICT>(s);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float r = a[k] * rs;                         // responsibility of slot k for channel c
            t.g1[k][c] = r * (xs[c] - t.mu[k][c]) * invs2;
            t.g2[k] += r * rme[k];
        }
    }
    t.like = expf(t.ll_sum);
    // Leave-one-out likelihood, iodine.py:321-328: (sum_j m_j p_j - m_k p_k) / (1 - m_k + 1e-5).  Where a mask saturates
    // (m_k -> 1) this is a cancellation divided by 1e-5: the value IS rounding noise amplified 1e5 x, in the reference too.  The
    // only defensible target is the reference's own sequence of rounded fp32 operations - product, sequential sum over the
    // slots (torch.sum over dim 1), difference with the SAME rounded product, denominator (1 - m) + 1e-5.  hipcc's default fp
    // contraction fuses m_k * p_k into the sum / the difference (an EXACT product where the reference has a rounded one: the
    // numerator then goes negative by an ulp where the reference's is exactly 0, i.e. -0.2 instead of 0 after the division) and
    // does so differently in the three kernels that inline this function.  HIP's __fmul_rn / __fadd_rn do NOT help: they are
    // plain operators in a header compiled with contraction on, and LLVM fuses them after inlining.  What helps is the
    // `fp contract(off)` pragma at the top of this function with the arithmetic written as plain operators HERE.  Found by
    // tests/test_gpu_trained_weights.py (sharpened masks): pass 2 of the split first refinement layer computed -101 sigma at a
    // pixel where pass 1's statistics had seen 0, and the ELBO of the following iterations was off by 8e-4 relative.
    float prod[K];
    t.mix = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        t.pk[k] = expf(lsum[k]);
        prod[k] = t.m[k] * t.pk[k];
        t.mix = k == 0 ? prod[0] : t.mix + prod[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) t.loo[k] = (t.mix - prod[k]) / ((1.f - t.m[k]) + 1e-5f);
}

template <int K, bool STRICT = false>
IOD_DEVINL void pixel_terms(const float4 xv, const float4* __restrict__ dec, size_t slot_stride, size_t p,
                            float inv2s2, float invs2, float lconst, PixelTerms<K>& t)
{
    float4 dv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) dv[k] = dec[(size_t)k * slot_stride + p];
    pixel_terms_core<K, STRICT>(xv, dv, inv2s2, invs2, lconst, t);
}
