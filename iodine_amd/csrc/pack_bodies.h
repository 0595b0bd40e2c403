// Element bodies of the split-fp16 weight packs, shared by the one-tensor kernels (kernels_conv.hip, kernels_convws.hip) and the batched
// form (kernels_pack.hip: every pack of iodine_set_params in two launches).  One definition each: the batched and the one-tensor form
// write the same bits.
#pragma once
#include "common.h"

// element idx of the weight-stationary register layout (kernels_convws.hip): [cout group][chunk of 32 cin][tap][hi/lo][lane][8 fp16]
IOD_DEVINL _Float16 pack_ws_element(const float* __restrict__ src, int C, int tflip, float scale, size_t idx)
{
    const int nchunk = C / 32;
    const int e = idx & 7;
    size_t r = idx >> 3;
    const int lane = r & 63; r >>= 6;
    const int hl = r & 1; r >>= 1;
    const int tap = r % 9; r /= 9;
    const int c = r % nchunk;
    const int cg = (int)(r / nchunk);
    const int co = 16 * cg + (lane & 15), ci = 32 * c + 8 * (lane >> 4) + e;
    // forward: W[co][ci][tap]; data gradient: the transposed conv, W[ci][co][8 - tap] (roles of the channel axes swapped)
    float v = tflip ? src[((size_t)ci * C + co) * 9 + (8 - tap)] : src[((size_t)co * C + ci) * 9 + tap];
    v *= scale;
    const _Float16 hi = (_Float16)v;
    return hl == 0 ? hi : (_Float16)(v - (float)hi);
}
IOD_DEVINL size_t pack_ws_total(int C) { return (size_t)(C / 16) * (C / 32) * 9 * 2 * 64 * 8; }

// element idx of the LDS-tile layout (kernels_conv.hip): [chunk of 16 cin][tap][hi/lo][k half][cout][8 fp16]
IOD_DEVINL _Float16 pack_f16_element(const float* __restrict__ src, int O, int I, int cout, int tflip, float scale, size_t idx)
{
    const int e = idx & 7;
    size_t r = idx >> 3;
    const int co = r % cout; r /= cout;
    const int kh = r & 1; r >>= 1;
    const int term = r & 1; r >>= 1;
    const int tap = r % 9;
    const int chunk = r / 9;
    const int ci = chunk * 16 + kh * 8 + e;
    float v = 0.f;
    if (!tflip) { if (ci < I && co < O) v = src[((size_t)co * I + ci) * 9 + tap]; }
    else if (tflip == 1) { if (ci < O && co < I) v = src[((size_t)ci * I + co) * 9 + (8 - tap)]; }
    else { if (ci < O && co < I) v = src[((size_t)ci * I + co) * 9 + tap]; }
    v *= scale;
    const _Float16 hi = (_Float16)v;
    return term == 0 ? hi : (_Float16)(v - (float)hi);
}
IOD_DEVINL size_t pack_f16_total(int cin, int cout) { return (size_t)(cin / 16) * 9 * 2 * 2 * cout * 8; }

// element idx of the fused first refinement layer's operand (kernels_refl0.hip): [O / 16][9 taps][hi / lo][64 lanes] x 4 fp16
IOD_DEVINL _Float16 pack_l0_element(const float* __restrict__ w, int CINW, float scale, size_t idx)
{
    const int j = idx & 3, lane = (idx >> 2) & 63, hl = (idx >> 8) & 1, tap = (int)((idx >> 9) % 9), g = (int)((idx >> 9) / 9);
    const int co = 16 * g + (lane & 15), ci = 4 * (lane >> 4) + j;
    const float v = ci < CINW ? w[((size_t)co * CINW + ci) * 9 + tap] * scale : 0.f;
    const _Float16 hi = (_Float16)v;
    return hl == 0 ? hi : (_Float16)(v - (float)hi);
}
IOD_DEVINL size_t pack_l0_total(int O) { return (size_t)(O / 16) * 9 * 2 * 64 * 4; }

// element idx of the output conv's GEMM-form operand (kernels_conv.hip: pack_dec_out_gemm_kernel)
IOD_DEVINL _Float16 pack_out_gemm_element(const float* __restrict__ w, int C, float scale, size_t idx)
{
    const int e = idx & 7;
    int r = (int)(idx >> 3);
    const int n = r & 63; r >>= 6;
    const int kh = r & 1; r >>= 1;
    const int term = r & 1; r >>= 1;
    const int ci = r * 16 + kh * 8 + e;
    float v = 0.f;
    if (n < 36) v = w[((size_t)(n & 3) * C + ci) * 9 + (n >> 2)] * scale;
    const _Float16 hi = (_Float16)v;
    return term == 0 ? hi : (_Float16)(v - (float)hi);
}
IOD_DEVINL size_t pack_out_gemm_total(int C) { return (size_t)(C / 16) * 2 * 2 * 64 * 8; }

// element idx of the output conv's data-gradient operand (kernels_out.hip: pack_dec_out_dgrad_kernel)
IOD_DEVINL _Float16 pack_out_dgrad_element(const float* __restrict__ w, int C, float scale, size_t idx)
{
    const int e = idx & 7;
    int r = (int)(idx >> 3);
    const int c = r % C; r /= C;
    const int kh = r & 1; r >>= 1;
    const int term = r & 1; r >>= 1;
    const int k = r * 16 + kh * 8 + e, tap = k >> 2, o = k & 3;
    float v = 0.f;
    if (tap < 9) v = w[((size_t)o * C + c) * 9 + (8 - tap)] * scale;
    const _Float16 hi = (_Float16)v;
    return term == 0 ? hi : (_Float16)(v - (float)hi);
}
IOD_DEVINL size_t pack_out_dgrad_total(int C) { return (size_t)3 * 2 * 2 * C * 8; }

// max |w| of a tensor (block of 1024 threads) -> meta[0] = power-of-two scale with max * scale in [2^12, 2^13), meta[1] = 1 / scale
IOD_DEVINL void weight_scale_block(const float* __restrict__ w, int n, float* __restrict__ meta, float* s_red /*[16]*/)
{
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(w[i]));
    m = wave_max_f32(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = 0.f;
        for (int k = 0; k < 16; ++k) mx = fmaxf(mx, s_red[k]);
        int e = 0;
        const bool ok = mx > 0.f && isfinite(mx);
        if (ok) frexpf(mx, &e);
        meta[0] = ok ? ldexpf(1.f, 13 - e) : 1.f;
        meta[1] = 1.f / meta[0];
    }
}
