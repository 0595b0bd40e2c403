// Split-fp16 (hi+lo, 3 MFMA, fp32 accumulate) kernels for the STRIDE-2 3x3 convs of the refinement network
// (RefinementNetwork, lib/modeling/iodine.py:446-503: REF.CONV_LAYERS x [conv k3 s2 p1 + ELU]) and their data /
// weight gradients.  gfx950 only.
//
// A stride-2 3x3 conv over a fine image (2Sc x 2Sc) is the sum of four stride-1 convs over the parity sub-images
//     X_sb[Y][X] = x[2Y+py][2X+px],   sb = (py,px)
// each with a SUBSET of the nine taps:  ky = 1 (dY = 0) for py = 0;  ky = 0 (dY = -1) or ky = 2 (dY = 0) for py = 1
// (same along x), i.e. 1 + 2 + 2 + 4 = 9 tap products in total - no MFMA work on structural zeros.  The forward
// kernel therefore reuses the stride-1 tile machinery of kernels_conv.hip (16x16 output tile, 4 waves x 64 px,
// 16-channel chunks split into fp16 hi/lo while being staged into LDS, block-local power-of-two scaling) and walks
// 4 * CIN/16 stages (sub-image, channel chunk) with 1, 2 or 4 taps each.  The data gradient is the mirror image:
// the four parity classes of fine pixels are four stride-1 convs of the coarse gradient with the same tap subsets
// (offsets 0/+1 instead of -1/0); one block computes one class of one 16x16 coarse tile.
#include "common.h"
#include <utility>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

IOD_DEVINL float elu1_fast_r(float v) { return v > 0.f ? v : __expf(v) - 1.f; }

template <class F, int... Is>
IOD_DEVINL void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
IOD_DEVINL void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// tap slot j of parity class sb = py*2+px
__host__ __device__ constexpr int s2_ntaps(int sb) { return (1 + (sb >> 1)) * (1 + (sb & 1)); }
__host__ __device__ constexpr int s2_ky(int sb, int j) { return (sb >> 1) ? 2 * (j / (1 + (sb & 1))) : 1; }
__host__ __device__ constexpr int s2_kx(int sb, int j) { return (sb & 1) ? 2 * (j % (1 + (sb & 1))) : 1; }
// LDS tile offset (rows or columns) of a tap: forward tiles start at coarse -1 (k = 0 reads -1, else 0);
// gradient tiles start at coarse 0 (k = 0 reads +1, else 0)
__host__ __device__ constexpr int s2_off(int mode, int k) { return mode == 0 ? (k == 0 ? 0 : 1) : (k == 0 ? 1 : 0); }
// forward stage order: the 4-tap class first (its weights are the largest staging burst)
__host__ __device__ constexpr int s2_fwd_sb(int i) { return 3 - i; }

// F32 (conv_precision 0, round 5): the same stages with IEEE fp32 products on v_mfma_f32_32x32x2_f32 - a staged pixel is its 16 fp32
// channels (64 of the same 80 bytes), a weight slot (term, kh, co) holds the four fp32 weights of input channels 8 kh + 4 term .. + 3
// (launch_pack_conv_weights_s2f32: same byte count and staging as the split pack), lane (kh, li) feeds k-step (term, e) with channel
// 8 kh + 4 term + e of its pixel and of its output channel; no scales, one barrier less per stage.
template <int CIN_REAL, int CIN, int COUT, int MODE, int SBT, bool F32>
IOD_DEVINL void conv_s2_body(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                             const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                             int Sc, int tiles, int kdiv, unsigned char* smem_b, int bid)
{
    constexpr int NC16 = CIN / 16;
    constexpr int NSTAGE = MODE == 0 ? 4 * NC16 : NC16;
    constexpr int NT = COUT / 32;
    constexpr int HALO = 17, NPX = HALO * HALO;
    constexpr int PXS = 80;                              // bytes per staged pixel: 32 hi + 32 lo + 16 pad
    constexpr int IN_BYTES = (NPX + 1) * PXS;            // +1 pixel: dump slot for idle lanes
    constexpr int TAP_U4 = 4 * COUT;                     // uint4 per tap: [term hi/lo][kh][co]
    constexpr int NIN = (NPX * 4 + 255) / 256;           // 5 float4 per thread per stage
    constexpr int NWT = (TAP_U4 + 255) / 256;
    static_assert(TAP_U4 % 256 == 0 || TAP_U4 < 256, "weight staging assumes whole or partial single pass");

    unsigned char* s_in = smem_b;
    uint4* s_w = reinterpret_cast<uint4*>(smem_b + IN_BYTES);
    float* s_max = reinterpret_cast<float*>(smem_b + IN_BYTES + 4 * TAP_U4 * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int prow = li >> 4, pcol = li & 15;
    const int Sf = 2 * Sc;

    const int tx = bid % tiles; bid /= tiles;
    const int ty = bid % tiles;
    const int n = bid / tiles;

    const int cq = tid & 3;
    int goff[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int idx = tid + k * 256;
        const int px = idx >> 2;
        const int r = px / HALO, c = px % HALO;
        if (MODE == 0) {
            const int Y = ty * 16 - 1 + r, X = tx * 16 - 1 + c;
            const bool ok = idx < NPX * 4 && Y >= 0 && Y < Sc && X >= 0 && X < Sc;
            goff[k] = ok ? ((n * Sf + 2 * Y) * Sf + 2 * X) * CIN_REAL + cq * 4 : -1;
        } else {
            const int Y = ty * 16 + r, X = tx * 16 + c;
            const bool ok = idx < NPX * 4 && Y < Sc && X < Sc;
            goff[k] = ok ? ((n * Sc + Y) * Sc + X) * CIN_REAL + cq * 4 : -1;
        }
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    float4 rin[2][NIN];
    uint4 rw[4][NWT];

    auto stage_sb = [](int I) constexpr { return MODE == 0 ? s2_fwd_sb(I / NC16) : SBT; };
    auto stage_c16 = [](int I) constexpr { return MODE == 0 ? I % NC16 : I; };

    auto prefetch_in = [&](auto Ic, float4 (&r)[NIN]) {
        constexpr int I = decltype(Ic)::value;
        constexpr int sb = stage_sb(I), c16 = stage_c16(I);
        const int soff = MODE == 0 ? ((sb >> 1) * Sf + (sb & 1)) * CIN_REAL + c16 * 16 : c16 * 16;
        const bool chv = c16 * 16 + cq * 4 < CIN_REAL;      // channels past CIN_REAL (17 -> 20 of a 32-wide pad) read as 0
        const float* base = in + soff;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const bool ok = goff[k] >= 0 && chv;
            const float4 v = *reinterpret_cast<const float4*>(base + (ok ? goff[k] : -soff));
            r[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto prefetch_w = [&](auto Ic) {
        constexpr int I = decltype(Ic)::value;
        constexpr int sb = stage_sb(I), c16 = stage_c16(I);
#pragma unroll
        for (int j = 0; j < s2_ntaps(sb); ++j) {
            const int tap = s2_ky(sb, j) * 3 + s2_kx(sb, j);
            const uint4* wsrc = wpk + (size_t)(c16 * 9 + tap) * TAP_U4;
#pragma unroll
            for (int k = 0; k < NWT; ++k) {
                const int idx = tid + k * 256;
                rw[j][k] = idx < TAP_U4 ? wsrc[idx] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    float cur_scale = 1.f;
    auto commit = [&](auto Ic, const float4 (&r)[NIN]) -> float {
        constexpr int I = decltype(Ic)::value;
        constexpr int sb = stage_sb(I);
        if constexpr (F32) {
            __syncthreads();                                 // every wave is done reading the previous stage
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const int idx = tid + k * 256;
                const int px = idx < NPX * 4 ? idx >> 2 : NPX;
                *reinterpret_cast<float4*>(s_in + px * PXS + cq * 16) = r[k];
            }
#pragma unroll
            for (int j = 0; j < s2_ntaps(sb); ++j)
#pragma unroll
                for (int k = 0; k < NWT; ++k) {
                    const int idx = tid + k * 256;
                    if (idx < TAP_U4) s_w[j * TAP_U4 + idx] = rw[j][k];
                }
            __syncthreads();
            return 1.f;
        }
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const float4 v = r[k];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        m = wave_max_f32(m);
        if (lane == 0) s_max[wv] = m;
        __syncthreads();                                     // also: every wave is done reading the previous stage
        const float mb = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        const int e = (int)((__float_as_uint(mb) >> 23) & 0xffu) - 127;
        int se = 12 - e;
        se = se > 100 ? 100 : (se < -100 ? -100 : se);
        const float scale = (mb > 0.f && mb < 3.0e38f) ? __uint_as_float((unsigned)(127 + se) << 23) : 1.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int idx = tid + k * 256;
            const int px = idx < NPX * 4 ? idx >> 2 : NPX;
            float4 v = r[k];
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
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
        for (int j = 0; j < s2_ntaps(sb); ++j)
#pragma unroll
            for (int k = 0; k < NWT; ++k) {
                const int idx = tid + k * 256;
                if (idx < TAP_U4) s_w[j * TAP_U4 + idx] = rw[j][k];
            }
        __syncthreads();
        return scale;
    };
    auto rescale = [&](float new_scale) {
        if (new_scale != cur_scale) {                        // block-uniform; exact (powers of two)
            const float r = new_scale / cur_scale;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[mt][nt][q] *= r;
            cur_scale = new_scale;
        }
    };
    const unsigned char* a_base = s_in + ((4 * wv + prow) * HALO + pcol) * PXS + kh * (F32 ? 32 : 16);
    const uint4* b_base = s_w + kh * COUT + li;
    auto compute = [&](auto Ic) {
        constexpr int I = decltype(Ic)::value;
        constexpr int sb = stage_sb(I);
#pragma unroll
        for (int j = 0; j < s2_ntaps(sb); ++j) {
            const int aoff = (s2_off(MODE, s2_ky(sb, j)) * HALO + s2_off(MODE, s2_kx(sb, j))) * PXS;
            if constexpr (F32) {
                f32x4 av[2][2], bv[NT][2];                   // [.][term]
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    av[mt][0] = *reinterpret_cast<const f32x4*>(a_base + aoff + mt * 2 * HALO * PXS);
                    av[mt][1] = *reinterpret_cast<const f32x4*>(a_base + aoff + mt * 2 * HALO * PXS + 16);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const uint4 v0 = b_base[j * TAP_U4 + nt * 32], v1 = b_base[j * TAP_U4 + 2 * COUT + nt * 32];
                    __builtin_memcpy(&bv[nt][0], &v0, 16); __builtin_memcpy(&bv[nt][1], &v1, 16);
                }
#pragma unroll
                for (int term = 0; term < 2; ++term)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[nt][term][e], av[mt][term][e], acc[mt][nt], 0, 0, 0);
                continue;
            }
            f16x8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ah[mt] = *reinterpret_cast<const f16x8*>(a_base + aoff + mt * 2 * HALO * PXS);
                al[mt] = *reinterpret_cast<const f16x8*>(a_base + aoff + mt * 2 * HALO * PXS + 32);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const uint4 vh = b_base[j * TAP_U4 + nt * 32], vl = b_base[j * TAP_U4 + 2 * COUT + nt * 32];
                __builtin_memcpy(&bh[nt], &vh, 16); __builtin_memcpy(&bl[nt], &vl, 16);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], al[mt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nt], ah[mt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], ah[mt], acc[mt][nt], 0, 0, 0);
                }
        }
    };

    // fully unrolled stage schedule (straight-line code keeps the vmcnt waits counted): inputs two stages ahead in
    // two alternating register sets, weights one stage ahead
    using std::integral_constant;
    prefetch_in(integral_constant<int, 0>{}, rin[0]);
    prefetch_w(integral_constant<int, 0>{});
    if constexpr (NSTAGE > 1) prefetch_in(integral_constant<int, 1>{}, rin[1]);
    // data gradient: the ELU' operand of the whole tile in the whole-pixel layout of the epilogue (lane = 16-byte segment seg of
    // pixel pl of an instruction), requested behind the first two stages' inputs so that it is there long before it is used
    constexpr int E_SEGS = COUT / 4, E_PPI = 64 / E_SEGS, E_NEP = 32 / E_PPI;
    float4 axv[MODE == 1 ? 2 : 1][MODE == 1 ? E_NEP : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < E_NEP; ++j) {
                const int pq = j * E_PPI + lane / E_SEGS;
                const int Y = ty * 16 + 4 * wv + 2 * mt + (pq >> 4), X = tx * 16 + (pq & 15);
                axv[mt][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (Y < Sc && X < Sc)
                    axv[mt][j] = *reinterpret_cast<const float4*>(
                        aux + (size_t)((n * Sf + 2 * Y + (SBT >> 1)) * Sf + 2 * X + (SBT & 1)) * COUT + (lane % E_SEGS) * 4);
            }
    }
    static_for<NSTAGE>([&](auto Ic) {
        constexpr int I = decltype(Ic)::value;
        const float sc = commit(Ic, rin[I & 1]);
        if constexpr (!F32) { if (I == 0) cur_scale = sc; else rescale(sc); }
        if constexpr (I + 1 < NSTAGE) prefetch_w(integral_constant<int, (I + 1 < NSTAGE ? I + 1 : 0)>{});
        if constexpr (I + 2 < NSTAGE) prefetch_in(integral_constant<int, (I + 2 < NSTAGE ? I + 2 : 0)>{}, rin[I & 1]);
        compute(Ic);
    });

    const float inv_ws = F32 ? 1.f : wmeta[1] / cur_scale;
    // MFMAs are issued as (weights, activations): accumulator rows are channels, lane li = pixel li of a 32-pixel half tile,
    // registers 4g..4g+3 = channels 8g + 4kh .. +3.  Stored directly, every instruction writes 16 bytes per lane at the pixel
    // stride (32 partial cache lines); so each wave transposes its half tile through the (now free) staging buffers and
    // stores / multiplies WHOLE pixels: 16 lanes x 16 bytes = one pixel, 4 (C = 64) or 8 (C = 32) pixels per instruction
    // (data gradient: the ELU' operand was requested in that layout at the start - fetched only now it arrives too late:
    // 346 -> 387 us at cfg3).
    __syncthreads();                                     // every wave is done reading the last stage's LDS
    constexpr int EPD = COUT + 4;                        // dwords per transposed pixel (conflict-free float4 writes)
    constexpr int SEGS = COUT / 4, PPI = 64 / SEGS, NEP = 32 / PPI;
    static_assert(4 * 32 * EPD * 4 <= IN_BYTES + 4 * TAP_U4 * 16, "transposition regions fit the staging buffers");
    float* ep = reinterpret_cast<float*>(smem_b) + wv * 32 * EPD;
    const int seg = lane % SEGS, pl = lane / SEGS;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<f32x4*>(ep + li * EPD + nt * 32 + 8 * g4 + 4 * kh) =
                    f32x4{acc[mt][nt][4 * g4] * inv_ws, acc[mt][nt][4 * g4 + 1] * inv_ws, acc[mt][nt][4 * g4 + 2] * inv_ws,
                          acc[mt][nt][4 * g4 + 3] * inv_ws};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < NEP; ++j) {
            const int pq = j * PPI + pl;
            const int Y = ty * 16 + 4 * wv + 2 * mt + (pq >> 4);
            const int X = tx * 16 + (pq & 15);
            if (Y >= Sc || X >= Sc) continue;
            const f32x4 t = *reinterpret_cast<const f32x4*>(ep + pq * EPD + seg * 4);
            float4 v = make_float4(t.x, t.y, t.z, t.w);
            const int c0 = seg * 4;
            if constexpr (MODE == 1) {
                const float4 a4 = axv[mt][j];
                v.x *= elu1_grad_from_out(a4.x); v.y *= elu1_grad_from_out(a4.y);
                v.z *= elu1_grad_from_out(a4.z); v.w *= elu1_grad_from_out(a4.w);
                *reinterpret_cast<float4*>(out + (size_t)((n * Sf + 2 * Y + (SBT >> 1)) * Sf + 2 * X + (SBT & 1)) * COUT + c0) = v;
                continue;
            }
            float4* dst = reinterpret_cast<float4*>(out + (size_t)((n * Sc + Y) * Sc + X) * COUT + c0);
            if (!bias) { *dst = v; continue; }           // raw form (block-uniform): the per-image part of a split first layer
            const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (kdiv) {                                  // + the per-image map of the channels all slots of an image share
                const float4 mv = *reinterpret_cast<const float4*>(aux + (size_t)(((n / kdiv) * Sc + Y) * Sc + X) * COUT + c0);
                v.x += mv.x; v.y += mv.y; v.z += mv.z; v.w += mv.w;
            }
            if constexpr (F32) *dst = make_float4(elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w));      // (expm1f: the VALU is idle beside fp32 MFMA)
            else *dst = make_float4(elu1_fast_r(v.x), elu1_fast_r(v.y), elu1_fast_r(v.z), elu1_fast_r(v.w));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// MODE 0: in = fine [N][2Sc][2Sc][CIN_REAL] -> out = coarse [N][Sc][Sc][COUT], bias + ELU
// MODE 1: in = coarse gradient [N][Sc][Sc][CIN] -> out = fine [N][2Sc][2Sc][COUT] times ELU'(aux); grid.y = parity class
#ifndef S2_FWD_WAVES
#define S2_FWD_WAVES 2
#endif
template <int CIN_REAL, int CIN, int COUT, int MODE, bool F32 = false>
__global__ __launch_bounds__(256, MODE == 0 ? S2_FWD_WAVES : 2)
void conv3x3_s2_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                             const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                             int Sc, int tiles, int kdiv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s2[];
    if (MODE == 0) {
        conv_s2_body<CIN_REAL, CIN, COUT, 0, 0, F32>(in, wpk, wmeta, bias, aux, out, Sc, tiles, kdiv, smem_s2, (int)blockIdx.x);
    } else {
        // The four parity classes of a coarse tile read the SAME staged input: their blocks are made neighbours in dispatch order ON
        // ONE XCD (block b runs on XCD b % 8, observed; speed only), so that three of the four reads hit that XCD's L2 instead of
        // coming from HBM at different times (round 3: 1.42x the algorithmic bytes).  b = 8 (4 q + class) + xcd, tile = 8 q + xcd.
        const int b = (int)blockIdx.x, xcd = b & 7, r = b >> 3, tile = (r >> 2) * 8 + xcd;
        if (tile >= kdiv) return;                            // (kdiv carries the tile count in this mode; block-uniform)
        switch (r & 3) {                                     // heaviest class (4 taps) first
        case 3: conv_s2_body<CIN_REAL, CIN, COUT, 1, 0, F32>(in, wpk, wmeta, bias, aux, out, Sc, tiles, 0, smem_s2, tile); break;
        case 2: conv_s2_body<CIN_REAL, CIN, COUT, 1, 1, F32>(in, wpk, wmeta, bias, aux, out, Sc, tiles, 0, smem_s2, tile); break;
        case 1: conv_s2_body<CIN_REAL, CIN, COUT, 1, 2, F32>(in, wpk, wmeta, bias, aux, out, Sc, tiles, 0, smem_s2, tile); break;
        default: conv_s2_body<CIN_REAL, CIN, COUT, 1, 3, F32>(in, wpk, wmeta, bias, aux, out, Sc, tiles, 0, smem_s2, tile); break;
        }
    }
}

template <int CIN_REAL, int CIN, int COUT, int MODE, bool F32 = false>
hipError_t launch_s2_inst(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias,
                          const float* aux, float* out, int N, int Sc, int kdiv = 0)
{
    constexpr size_t lds = (size_t)(17 * 17 + 1) * 80 + (size_t)4 * 4 * COUT * 16 + 16;
    static std::atomic<unsigned> attr_devs{0};                             // devices this instance is configured on
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_s2_f16x3_kernel<CIN_REAL, CIN, COUT, MODE, F32>, (int)lds, attr_devs); e != hipSuccess) return e;
    const int tiles = (Sc + 15) / 16;
    const int ntiles = N * tiles * tiles;
    if (MODE == 1) kdiv = ntiles;                                          // data gradient: 1-D grid, see the kernel
    hipLaunchKernelGGL((conv3x3_s2_f16x3_kernel<CIN_REAL, CIN, COUT, MODE, F32>), dim3(MODE == 0 ? ntiles : ((ntiles + 7) / 8) * 32),
                       dim3(256), lds, st, in, reinterpret_cast<const uint4*>(wpk), wmeta, bias, aux, out, Sc, tiles, kdiv);
    return hipGetLastError();
}

}  // namespace

// Forward stride-2 conv + bias + ELU.  S = fine (input) size, even; cin_real = floats per input pixel (20 for the
// 17-channel encoding, packed with cin_pad = 32), wpk = launch_pack_conv_weights_f16(.., cin_pad, cout, tflip 0).
// Split first layer (the encoding's channels that all slots of an image share are convolved once per image):
//   bias == nullptr          raw result, no bias / ELU (the per-image part; cin_real 8, packed with cin_pad 16)
//   addmap != nullptr, kdiv  out = ELU(conv + bias + addmap[n / kdiv])   (the per-slot part; cin_real 12, cin_pad 16)
// f32 = 1: exact fp32 MFMA form (wpk = launch_pack_conv_weights_s2f32 with the same cin_pad / cout / tflip; wmeta unused).
hipError_t launch_conv3x3_s2_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias,
                                   float* out, int N, int S, int cin_real, int cout, const float* addmap, int kdiv, int f32)
{
    IOD_XSKIP(512);
    if (S % 2 != 0 || S < 2 || (addmap != nullptr) != (kdiv > 0)) return hipErrorInvalidValue;
#define S2F_CASE(CR, CP, CO) \
    if (cin_real == CR && cout == CO) return f32 ? launch_s2_inst<CR, CP, CO, 0, true>(st, in, wpk, wmeta, bias, addmap, out, N, S / 2, kdiv) \
                                                 : launch_s2_inst<CR, CP, CO, 0>(st, in, wpk, wmeta, bias, addmap, out, N, S / 2, kdiv);
    S2F_CASE(20, 32, 64) S2F_CASE(64, 64, 64) S2F_CASE(20, 32, 32) S2F_CASE(32, 32, 32)
    S2F_CASE(12, 16, 64) S2F_CASE(8, 16, 64) S2F_CASE(12, 16, 32) S2F_CASE(8, 16, 32)
#undef S2F_CASE
    return hipErrorInvalidValue;
}

namespace {
// fp32 weights in the LDS-tile layout of conv_s2_body<.., F32 = true>: [chunk of 16 cin][tap][term][k half][cout] x 4 floats,
// element e of slot (term, kh, co) = input channel 16 chunk + 8 kh + 4 term + e (tflip as in pack_f16_element)
__global__ void pack_conv_weights_s2f32_kernel(const float* __restrict__ src, int O, int I, int cin, int cout, int tflip, float* __restrict__ dst)
{
    const size_t total = (size_t)(cin / 16) * 9 * 2 * 2 * cout * 4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        size_t r = idx >> 2;
        const int co = r % cout; r /= cout;
        const int kh = r & 1; r >>= 1;
        const int term = r & 1; r >>= 1;
        const int tap = r % 9;
        const int chunk = (int)(r / 9);
        const int ci = chunk * 16 + kh * 8 + term * 4 + e;
        float v = 0.f;
        if (!tflip) { if (ci < I && co < O) v = src[((size_t)co * I + ci) * 9 + tap]; }
        else if (tflip == 1) { if (ci < O && co < I) v = src[((size_t)ci * I + co) * 9 + (8 - tap)]; }
        else { if (ci < O && co < I) v = src[((size_t)ci * I + co) * 9 + tap]; }
        dst[idx] = v;
    }
}
}  // namespace

hipError_t launch_pack_conv_weights_s2f32(hipStream_t st, const float* src, int O, int I, int cin, int cout, int tflip, void* dst)
{
    const size_t total = (size_t)(cin / 16) * 9 * 2 * 2 * cout * 4;
    hipLaunchKernelGGL(pack_conv_weights_s2f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, O, I, cin, cout, tflip,
                       (float*)dst);
    return hipGetLastError();
}

// Data gradient of the stride-2 conv times ELU'(aux): d = coarse [N][S/2][S/2][cout_conv] -> out = fine [N][S][S][cin_conv];
// wpk = launch_pack_conv_weights_f16(w, O, I, cin_pad = O(conv out), cout = I(conv in), tflip 2).
hipError_t launch_conv3x3_s2_dgrad_f16x3(hipStream_t st, const float* d, const void* wpk, const float* wmeta, const float* aux,
                                         float* out, int N, int S, int c, int f32)
{
    IOD_XSKIP(1024);
    if (S % 2 != 0 || S < 2) return hipErrorInvalidValue;
    if (c == 64) return f32 ? launch_s2_inst<64, 64, 64, 1, true>(st, d, wpk, wmeta, nullptr, aux, out, N, S / 2)
                            : launch_s2_inst<64, 64, 64, 1>(st, d, wpk, wmeta, nullptr, aux, out, N, S / 2);
    if (c == 32) return f32 ? launch_s2_inst<32, 32, 32, 1, true>(st, d, wpk, wmeta, nullptr, aux, out, N, S / 2)
                            : launch_s2_inst<32, 32, 32, 1>(st, d, wpk, wmeta, nullptr, aux, out, N, S / 2);
    return hipErrorInvalidValue;
}

// =========================================================================================
// Weight gradient of the stride-2 conv, split-fp16:  dW[tap][ci][co] = sum_px a[2oy+ky-1][2ox+kx-1][ci] * d[oy][ox][co]
//   M = ci, N = co, K = coarse pixels (16 per MFMA = one coarse tile row), as in the stride-1 kernel of
//   kernels_train.hip; the fine input is staged TRANSPOSED (one fp16 plane per channel) and DE-INTERLEAVED by column
//   parity, so that the K-consecutive operand of every tap is contiguous:
//     kx = 1 -> even plane at X,  kx = 2 -> odd plane at X,  kx = 0 -> odd plane at X-1 (register shift, v_alignbit)
//   plane row (20 dwords, 2 px each): [3] = odd (X=-2,-1) | [4..11] odd X=0..15 | [12..19] even X=0..15
//   tile = TH x 16 coarse pixels (fine halo 2TH+1 rows), persistent blocks, one partial tile set per (block, K-split).
// =========================================================================================
namespace {

template <int CI_REAL, int CI, int NCO, int TH>
__global__ __launch_bounds__(256, 2)
void conv3x3_s2_wgrad_f16x3_kernel(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ part,
                                   float* __restrict__ part_b, int Sc, int ntiles, int tiles_x, int tiles_y,
                                   const float* __restrict__ a2, int kdiv)
{
    constexpr int MT = CI / 32, NTT = NCO / 32, KS = 4 / (MT * NTT);
    static_assert(TH % KS == 0, "tile rows must split evenly over the K-split waves");
    constexpr int RW = TH / KS;
    constexpr int HR = 2 * TH + 1;
    constexpr int APL = HR * 20, DPL = TH * 8 + 4;       // dwords per channel plane (APL/4, DPL/4 odd: conflict-free b128)
    static_assert((APL / 4) % 2 == 1 && (DPL / 4) % 2 == 1, "plane strides");
    constexpr int A4 = CI_REAL / 4, D4 = NCO / 4;
    constexpr int NA_UNITS = HR * 17 * A4, ND_UNITS = TH * 8 * D4;
    constexpr int NAU = (NA_UNITS + 255) / 256, NDU = (ND_UNITS + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned smem_w2[];
    unsigned* s_a = smem_w2;                             // [2][CI][APL]
    unsigned* s_d = smem_w2 + 2 * CI * APL;              // [2][NCO][DPL]
    float* s_max = reinterpret_cast<float*>(s_d + 2 * NCO * DPL);     // [8]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int mi = wv % MT, ni = (wv / MT) % NTT, ks = wv / (MT * NTT);
    const int ci = mi * 32 + li, co = ni * 32 + li;
    const int Sf = 2 * Sc;

    if (CI_REAL < CI) {                                  // planes of the pad channels stay zero for the whole kernel
        for (int i = tid; i < 2 * CI * APL; i += 256) s_a[i] = 0u;
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float sa = 1.f, sd = 1.f, acc_prod = 1.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        // split first layer (a2 != nullptr, CI_REAL == 20): channels 0..11 from the per-slot tensor a[n] (12 floats per pixel),
        // channels 12..19 from the per-image tensor a2[n / kdiv] (8 floats per pixel)
        const float* a_n = a + (size_t)n * Sf * Sf * (a2 ? 12 : CI_REAL);
        const float* a2_n = a2 ? a2 + (size_t)(n / kdiv) * Sf * Sf * 8 : nullptr;
        const float* d_n = d + (size_t)n * Sc * Sc * NCO;

        float4 ra[NAU][2], rd[NDU][2];
        float ma = 0.f, md = 0.f;
#pragma unroll
        for (int k = 0; k < NAU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % A4, tt = u / A4, slot = tt % 17, row = tt / 17;
            const int par = slot <= 8 ? 1 : 0;
            const int j = slot == 0 ? -1 : (slot <= 8 ? slot - 1 : slot - 9);
            const int fy = 2 * ty * TH - 1 + row;
            const int fx0 = 2 * (tx * 16 + 2 * j) + par, fx1 = fx0 + 2;
            ra[k][0] = ra[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < NA_UNITS && fy >= 0 && fy < Sf) {
                const bool sh = a2_n && c4 >= 3;
                const float* src = sh ? a2_n + (c4 - 3) * 4 : a_n + c4 * 4;
                const int pst = a2_n ? (sh ? 8 : 12) : CI_REAL;
                if (fx0 >= 0 && fx0 < Sf) ra[k][0] = *reinterpret_cast<const float4*>(src + ((size_t)fy * Sf + fx0) * pst);
                if (fx1 >= 0 && fx1 < Sf) ra[k][1] = *reinterpret_cast<const float4*>(src + ((size_t)fy * Sf + fx1) * pst);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                ma = fmaxf(ma, fmaxf(fmaxf(fabsf(ra[k][q].x), fabsf(ra[k][q].y)), fmaxf(fabsf(ra[k][q].z), fabsf(ra[k][q].w))));
        }
#pragma unroll
        for (int k = 0; k < NDU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % D4, tt = u / D4, p = tt % 8, row = tt / 8;
            const int gy = ty * TH + row, gx = tx * 16 + 2 * p;
            rd[k][0] = rd[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < ND_UNITS && gy < Sc) {
                if (gx < Sc) rd[k][0] = *reinterpret_cast<const float4*>(d_n + ((size_t)gy * Sc + gx) * NCO + c4 * 4);
                if (gx + 1 < Sc) rd[k][1] = *reinterpret_cast<const float4*>(d_n + ((size_t)gy * Sc + gx + 1) * NCO + c4 * 4);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                md = fmaxf(md, fmaxf(fmaxf(fabsf(rd[k][q].x), fabsf(rd[k][q].y)), fmaxf(fabsf(rd[k][q].z), fabsf(rd[k][q].w))));
                bsum.x += rd[k][q].x; bsum.y += rd[k][q].y; bsum.z += rd[k][q].z; bsum.w += rd[k][q].w;
            }
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
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[tp][q] *= r;
            acc_prod = prod;
        }

#pragma unroll
        for (int k = 0; k < NAU; ++k) {
            const int u = tid + k * 256;
            if (u < NA_UNITS) {
                const int c4 = u % A4, tt = u / A4, slot = tt % 17, row = tt / 17;
                const int rot = c4 & 3;
                const float4 q0 = rot4(ra[k][0], rot), q1 = rot4(ra[k][1], rot);
                const float x0[4] = {q0.x, q0.y, q0.z, q0.w}, x1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned lo;
                    const unsigned hi = pack_hi_lo(x0[e] * sa, x1[e] * sa, lo);
                    const int ch = c4 * 4 + ((e + rot) & 3);
                    s_a[(0 * CI + ch) * APL + row * 20 + slot + 3] = hi;
                    s_a[(1 * CI + ch) * APL + row * 20 + slot + 3] = lo;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NDU; ++k) {
            const int u = tid + k * 256;
            if (u < ND_UNITS) {
                const int c4 = u % D4, tt = u / D4, p = tt % 8, row = tt / 8;
                const int rot = c4 & 3;
                const float4 q0 = rot4(rd[k][0], rot), q1 = rot4(rd[k][1], rot);
                const float x0[4] = {q0.x, q0.y, q0.z, q0.w}, x1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned lo;
                    const unsigned hi = pack_hi_lo(x0[e] * sd, x1[e] * sd, lo);
                    const int ch = c4 * 4 + ((e + rot) & 3);
                    s_d[(0 * NCO + ch) * DPL + row * 8 + p] = hi;
                    s_d[(1 * NCO + ch) * DPL + row * 8 + p] = lo;
                }
            }
        }
        __syncthreads();

#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = ks * RW + rr;
            h16x8 bh, bl;
            {
                const uint4 vb_h = *reinterpret_cast<const uint4*>(s_d + (0 * NCO + co) * DPL + r * 8 + 4 * kh);
                const uint4 vb_l = *reinterpret_cast<const uint4*>(s_d + (1 * NCO + co) * DPL + r * 8 + 4 * kh);
                __builtin_memcpy(&bh, &vb_h, 16); __builtin_memcpy(&bl, &vb_l, 16);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                h16x8 A[2][3];                               // [term][kx]
#pragma unroll
                for (int term = 0; term < 2; ++term) {
                    const unsigned* pl = s_a + (term * CI + ci) * APL + (2 * r + ky) * 20;
                    const uint4 vo = *reinterpret_cast<const uint4*>(pl + 4 + 4 * kh);         // odd plane X = 8kh .. 8kh+7
                    const uint4 ve = *reinterpret_cast<const uint4*>(pl + 12 + 4 * kh);        // even plane
                    const unsigned prev = pl[3 + 4 * kh] & 0xffff0000u;                        // odd X = 8kh-1 in the high half
                    uint4 m1;
                    m1.x = __builtin_amdgcn_alignbit(vo.x, prev, 16);
                    m1.y = __builtin_amdgcn_alignbit(vo.y, vo.x, 16);
                    m1.z = __builtin_amdgcn_alignbit(vo.z, vo.y, 16);
                    m1.w = __builtin_amdgcn_alignbit(vo.w, vo.z, 16);
                    __builtin_memcpy(&A[term][0], &m1, 16);
                    __builtin_memcpy(&A[term][1], &ve, 16);
                    __builtin_memcpy(&A[term][2], &vo, 16);
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int tap = ky * 3 + kx;
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][kx], bh, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][kx], bl, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][kx], bh, acc[tap], 0, 0, 0);
                }
            }
        }
    }

    const float inv = 1.f / acc_prod;
    float* pw = part + ((size_t)(blockIdx.x * KS + ks) * 9) * CI * NCO;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            pw[((size_t)tap * CI + cr) * NCO + ni * 32 + li] = acc[tap][r] * inv;
        }
    __syncthreads();
    float4* s_red = reinterpret_cast<float4*>(smem_w2);
    s_red[tid] = bsum;
    __syncthreads();
    if (tid < D4) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = tid; j < 256; j += D4) { const float4 v = s_red[j]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
        *reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * NCO + tid * 4) = t4;
    }
}

// Exact-fp32 form (conv_precision 0, round 5): dW on v_mfma_f32_32x32x2_f32 with K = two neighbouring coarse pixels.  Both operands are
// staged as fp32 CHANNEL PLANES (one ds_write_b32 per element, plane strides odd: the 16 channel quads x 4 pixels a wave writes per
// instruction hit 64 distinct banks), the fine input un-de-interleaved: lane (kh, li) reads channel li at fine column 4 s + 2 kh + kx of
// halo row 2 r + ky - one ds_read_b32 with an immediate offset per MFMA operand, 80 reads for the 72 MFMAs (4608 matrix-pipe cycles) of a
// tile row, so nothing but the matrix pipe matters.  The next tile's loads are in flight under the current tile's MFMAs; same partial-tile
// / bias-partial interface as the split kernel above.
template <int CI_REAL, int CI, int NCO, int TH>
__global__ __launch_bounds__(256, 2)
void conv3x3_s2_wgrad_f32_kernel(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ part,
                                 float* __restrict__ part_b, int Sc, int ntiles, int tiles_x, int tiles_y,
                                 const float* __restrict__ a2, int kdiv)
{
    constexpr int MT = CI / 32, NTT = NCO / 32, KS = 4 / (MT * NTT);
    static_assert(TH % KS == 0, "tile rows must split evenly over the K-split waves");
    constexpr int RW = TH / KS;
    constexpr int HR = 2 * TH + 1, AWD = 33;
    constexpr int APL = HR * AWD, DPL = TH * 16 + 1;     // floats per channel plane, both odd
    static_assert(APL % 2 == 1 && DPL % 2 == 1, "plane strides");
    constexpr int A4 = CI_REAL / 4, D4 = NCO / 4;
    constexpr int NA_UNITS = HR * AWD * A4, ND_UNITS = TH * 16 * D4;
    constexpr int NAU = (NA_UNITS + 255) / 256, NDU = (ND_UNITS + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned smem_w2[];
    float* s_a = reinterpret_cast<float*>(smem_w2);      // [CI][APL]
    float* s_d = s_a + CI * APL;                         // [NCO][DPL]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int mi = wv % MT, ni = (wv / MT) % NTT, ks = wv / (MT * NTT);
    const int Sf = 2 * Sc;

    if (CI_REAL < CI) {                                  // planes of the pad channels stay zero for the whole kernel
        for (int i = CI_REAL * APL + tid; i < CI * APL; i += 256) s_a[i] = 0.f;
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 ra[NAU], rd[NDU];
    auto load_tile = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        // split first layer (a2 != nullptr, CI_REAL == 20): channels 0..11 from the per-slot tensor a[n] (12 floats per pixel),
        // channels 12..19 from the per-image tensor a2[n / kdiv] (8 floats per pixel)
        const float* a_n = a + (size_t)n * Sf * Sf * (a2 ? 12 : CI_REAL);
        const float* a2_n = a2 ? a2 + (size_t)(n / kdiv) * Sf * Sf * 8 : nullptr;
        const float* d_n = d + (size_t)n * Sc * Sc * NCO;
#pragma unroll
        for (int k = 0; k < NAU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % A4, tt = u / A4, col = tt % AWD, row = tt / AWD;
            const int fy = 2 * ty * TH - 1 + row, fx = 32 * tx - 1 + col;
            ra[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < NA_UNITS && fy >= 0 && fy < Sf && fx >= 0 && fx < Sf) {
                const bool sh = a2_n && c4 >= 3;
                const float* src = sh ? a2_n + (c4 - 3) * 4 : a_n + c4 * 4;
                const int pst = a2_n ? (sh ? 8 : 12) : CI_REAL;
                ra[k] = *reinterpret_cast<const float4*>(src + ((size_t)fy * Sf + fx) * pst);
            }
        }
#pragma unroll
        for (int k = 0; k < NDU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % D4, tt = u / D4, p = tt % 16, row = tt / 16;
            const int gy = ty * TH + row, gx = tx * 16 + p;
            rd[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < ND_UNITS && gy < Sc && gx < Sc) rd[k] = *reinterpret_cast<const float4*>(d_n + ((size_t)gy * Sc + gx) * NCO + c4 * 4);
        }
    };

    const float* pa = s_a + (mi * 32 + li) * APL + 2 * kh;
    const float* pd = s_d + (ni * 32 + li) * DPL + kh;

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                   // every wave is done with the previous tile's planes
#pragma unroll
        for (int k = 0; k < NAU; ++k) {
            const int u = tid + k * 256;
            if (u < NA_UNITS) {
                const int c4 = u % A4, tt = u / A4;      // tt = row * AWD + col
                float* dst = s_a + (c4 * 4) * APL + tt;
                dst[0] = ra[k].x; dst[APL] = ra[k].y; dst[2 * APL] = ra[k].z; dst[3 * APL] = ra[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < NDU; ++k) {
            const int u = tid + k * 256;
            if (u < ND_UNITS) {
                const int c4 = u % D4, tt = u / D4;      // tt = row * 16 + p
                float* dst = s_d + (c4 * 4) * DPL + tt;
                dst[0] = rd[k].x; dst[DPL] = rd[k].y; dst[2 * DPL] = rd[k].z; dst[3 * DPL] = rd[k].w;
                bsum.x += rd[k].x; bsum.y += rd[k].y; bsum.z += rd[k].z; bsum.w += rd[k].w;
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + (int)gridDim.x);     // in flight under the MFMAs below

#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = ks * RW + rr;
            float bq[8];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) bq[s8] = pd[r * 16 + 2 * s8];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* prow = pa + (2 * r + ky) * AWD;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(prow[4 * s8 + kx], bq[s8], acc[ky * 3 + kx], 0, 0, 0);
            }
        }
    }

    float* pw = part + ((size_t)(blockIdx.x * KS + ks) * 9) * CI * NCO;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            pw[((size_t)tap * CI + cr) * NCO + ni * 32 + li] = acc[tap][r];
        }
    __syncthreads();
    float4* s_red = reinterpret_cast<float4*>(smem_w2);
    s_red[tid] = bsum;
    __syncthreads();
    if (tid < D4) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = tid; j < 256; j += D4) { const float4 v = s_red[j]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
        *reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * NCO + tid * 4) = t4;
    }
}

template <int CI_REAL, int CI, int NCO, int TH>
hipError_t launch_s2_wgrad32_inst(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N, int Sc,
                                  int* nparts, int* cipad, int* nbias_parts, const float* a2, int kdiv)
{
    constexpr int MT = CI / 32, NTT = NCO / 32, KS = 4 / (MT * NTT);
    constexpr size_t lds = (size_t)(CI * (2 * TH + 1) * 33 + NCO * (TH * 16 + 1)) * 4;
    static_assert(lds >= 256 * 16, "bias reduction reuses the planes");
    static std::atomic<unsigned> attr_devs{0};
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_s2_wgrad_f32_kernel<CI_REAL, CI, NCO, TH>, (int)lds, attr_devs); e != hipSuccess) return e;
    const int tiles_x = (Sc + 15) / 16, tiles_y = (Sc + TH - 1) / TH, ntiles = N * tiles_x * tiles_y;
    const int blocks = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL((conv3x3_s2_wgrad_f32_kernel<CI_REAL, CI, NCO, TH>), dim3(blocks), dim3(256), lds, st, a, d, part,
                       part_b, Sc, ntiles, tiles_x, tiles_y, a2, kdiv);
    *nparts = blocks * KS;
    *cipad = CI;
    *nbias_parts = blocks;
    return hipGetLastError();
}

template <int CI_REAL, int CI, int NCO, int TH>
hipError_t launch_s2_wgrad_inst(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N, int Sc,
                                int* nparts, int* cipad, int* nbias_parts, const float* a2, int kdiv)
{
    constexpr int MT = CI / 32, NTT = NCO / 32, KS = 4 / (MT * NTT);
    constexpr size_t lds = (size_t)(2 * CI * (2 * TH + 1) * 20 + 2 * NCO * (TH * 8 + 4)) * 4 + 32;
    static_assert(lds >= 256 * 16, "bias reduction reuses the planes");
    static std::atomic<unsigned> attr_devs{0};                             // devices this instance is configured on
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_s2_wgrad_f16x3_kernel<CI_REAL, CI, NCO, TH>, (int)lds, attr_devs); e != hipSuccess) return e;
    const int tiles_x = (Sc + 15) / 16, tiles_y = (Sc + TH - 1) / TH, ntiles = N * tiles_x * tiles_y;
    const int blocks = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL((conv3x3_s2_wgrad_f16x3_kernel<CI_REAL, CI, NCO, TH>), dim3(blocks), dim3(256), lds, st, a, d, part,
                       part_b, Sc, ntiles, tiles_x, tiles_y, a2, kdiv);
    *nparts = blocks * KS;
    *cipad = CI;
    *nbias_parts = blocks;
    return hipGetLastError();
}

}  // namespace

// part: nparts x [9][cipad][nco] partial tiles (reduce with launch_wgrad_reduce), part_b: nbias_parts x [nco] bias partials.
// S = fine (input) size, even.
// a2 / kdiv: split first layer - 12 per-slot channels from a, 8 per-image channels from a2[n / kdiv] (ci_real 20 only).
hipError_t launch_conv3x3_s2_wgrad_f16x3(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N,
                                         int S, int ci_real, int nco, int* nparts, int* cipad, int* nbias_parts,
                                         const float* a2, int kdiv, int f32)
{
    IOD_XSKIP(1024);
    if (S % 2 != 0 || S < 2 || (a2 && (ci_real != 20 || kdiv < 1))) return hipErrorInvalidValue;
#define S2W_CASE(CR, CP, CO, TH) \
    if (ci_real == CR && nco == CO) \
        return f32 ? launch_s2_wgrad32_inst<CR, CP, CO, TH>(st, a, d, part, part_b, N, S / 2, nparts, cipad, nbias_parts, a2, kdiv) \
                   : launch_s2_wgrad_inst<CR, CP, CO, TH>(st, a, d, part, part_b, N, S / 2, nparts, cipad, nbias_parts, a2, kdiv);
    S2W_CASE(64, 64, 64, 2) S2W_CASE(20, 32, 64, 4) S2W_CASE(32, 32, 32, 4) S2W_CASE(20, 32, 32, 4)
#undef S2W_CASE
    return hipErrorInvalidValue;
}

// =========================================================================================
// Split first layer of the refinement network.  The 17-channel encoding (IODINE.get_input_encoding, iodine.py:243-343) has
// 6 channels that every slot of an image shares (image rgb, LN(pixel likelihood), the two coordinate channels) and 11 that
// differ; a conv is linear in its input channels, so layer 0 = ELU(bias + conv_11(per slot) + conv_6(per image)): the shared
// part is convolved once per image, the per-slot tensor shrinks from 20 to 12 floats per pixel and from two 16-channel chunks
// (one of them 3/4 padding) to one.  Internal channel order: 0..11 per-slot (11 = pad), 12..19 per-image (18, 19 = pad).
// =========================================================================================
namespace {
__device__ __constant__ int c_ref_split_map[20] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, -1, 0, 1, 2, 13, 15, 16, -1, -1};

// w [O][17][9] -> w_slot [O][12][9], w_sh [O][8][9] (pad channels zero)
__global__ void ref_split_weights_kernel(const float* __restrict__ w, int O, float* __restrict__ w_slot, float* __restrict__ w_sh)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * 20 * 9) return;
    const int t = idx % 9, i = (idx / 9) % 20, o = idx / 180;
    const int r = c_ref_split_map[i];
    const float v = r >= 0 ? w[((size_t)o * 17 + r) * 9 + t] : 0.f;
    if (i < 12) w_slot[((size_t)o * 12 + i) * 9 + t] = v;
    else w_sh[((size_t)o * 8 + (i - 12)) * 9 + t] = v;
}

// gw [O][17][9] += g20 [O][20][9] in the internal channel order
__global__ void ref_unsplit_grad_kernel(const float* __restrict__ g20, int O, float* __restrict__ gw)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * 20 * 9) return;
    const int t = idx % 9, i = (idx / 9) % 20, o = idx / 180;
    const int r = c_ref_split_map[i];
    if (r >= 0) gw[((size_t)o * 17 + r) * 9 + t] += g20[idx];
}

// the 20-channel encoding (17 + 3 zero pad, reference channel order) from the split tensors - debug / test read-back only
__global__ void enc_join_kernel(const float* __restrict__ enck, const float* __restrict__ encs, float* __restrict__ enc, int K, int P,
                                size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % 20);
    const size_t px = idx / 20, n = px / P, p = px % P;
    float v = 0.f;
    if (c < 17) {
        int i = 0;
        for (int q = 0; q < 20; ++q) if (c_ref_split_map[q] == c) i = q;
        v = i < 12 ? enck[px * 12 + i] : encs[((n / K) * P + p) * 8 + (i - 12)];
    }
    enc[idx] = v;
}
}  // namespace

// ---- ARCH.ENCODING subsets (iodine.py:277-340 tests `in self.encodings` entry by entry): the kernels always work on the 17
// internal channels; a first-layer weight with fewer input channels is EXPANDED (absent channels get zero weights, so whatever
// pixel_pass2 writes there does not matter) and its gradient is GATHERED back ----------------------------------------------
namespace {
struct EncMap { int c[17]; };
// w [O][n_in][kk] -> w17 [O][17][kk]   (kk = taps: 9, or 25 / 49 on the generic path)
__global__ void enc_expand_weights_kernel(const float* __restrict__ w, int O, int n_in, int kk, EncMap map, float* __restrict__ w17)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * 17 * kk) return;
    const int t = idx % kk, c = (idx / kk) % 17, o = idx / (17 * kk);
    float v = 0.f;
    for (int j = 0; j < n_in; ++j) if (map.c[j] == c) v = w[((size_t)o * n_in + j) * kk + t];
    w17[idx] = v;
}
// gw [O][n_in][kk] += g17 [O][17][kk] at the present channels
__global__ void enc_gather_grad_kernel(const float* __restrict__ g17, int O, int n_in, int kk, EncMap map, float* __restrict__ gw)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * n_in * kk) return;
    const int t = idx % kk, j = (idx / kk) % n_in, o = idx / (kk * n_in);
    gw[idx] += g17[((size_t)o * 17 + map.c[j]) * kk + t];
}
}  // namespace

hipError_t launch_enc_expand_weights(hipStream_t st, const float* w, int O, int n_in, const int* map17, float* w17, int kk)
{
    EncMap m;
    for (int j = 0; j < 17; ++j) m.c[j] = j < n_in ? map17[j] : -1;
    hipLaunchKernelGGL(enc_expand_weights_kernel, dim3((O * 17 * kk + 255) / 256), dim3(256), 0, st, w, O, n_in, kk, m, w17);
    return hipGetLastError();
}

hipError_t launch_enc_gather_grad(hipStream_t st, const float* g17, int O, int n_in, const int* map17, float* gw, int kk)
{
    EncMap m;
    for (int j = 0; j < 17; ++j) m.c[j] = j < n_in ? map17[j] : -1;
    hipLaunchKernelGGL(enc_gather_grad_kernel, dim3((O * n_in * kk + 255) / 256), dim3(256), 0, st, g17, O, n_in, kk, m, gw);
    return hipGetLastError();
}

hipError_t launch_ref_split_weights(hipStream_t st, const float* w, int O, float* w_slot, float* w_sh)
{
    hipLaunchKernelGGL(ref_split_weights_kernel, dim3((O * 180 + 255) / 256), dim3(256), 0, st, w, O, w_slot, w_sh);
    return hipGetLastError();
}

hipError_t launch_ref_unsplit_grad(hipStream_t st, const float* g20, int O, float* gw)
{
    hipLaunchKernelGGL(ref_unsplit_grad_kernel, dim3((O * 180 + 255) / 256), dim3(256), 0, st, g20, O, gw);
    return hipGetLastError();
}

hipError_t launch_enc_join(hipStream_t st, const float* enck, const float* encs, float* enc, int N, int K, int P)
{
    const size_t total = (size_t)N * P * 20;
    hipLaunchKernelGGL(enc_join_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, enck, encs, enc, K, P, total);
    return hipGetLastError();
}
