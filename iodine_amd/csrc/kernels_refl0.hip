// get_input_encoding + the first layer of the refinement network in ONE kernel (round 4).  gfx950 only.
//
// Reference: IODINE.get_input_encoding (lib/modeling/iodine.py:243-343: the 17 image-shaped channels, 5-D layer-norm :385-394) followed by
// RefinementNetwork.mlc.layers[0] (conv k3 s2 p1 17 -> C + ELU, iodine.py:459,480).  Until now three launches per refinement iteration:
// pixel_pass2 wrote the encoding (176 + 17 MB at cfg3, split into the 11 channels that differ between the slots of an image and the 6 they
// share), a stride-2 conv reduced the shared part per image, a second one the per-slot part: 0.064 + 0.020 + 0.146 ms for
// 0.06 GB of real input (the decoder output) and 0.23 GB of output.  Here a persistent 448-thread block takes one 2 x 16 output tile of ONE
// IMAGE for ALL K slots per unit, two roles, LDS planes double-buffered:
//   * producers (waves 4-6, 165 threads = the 5 x 33 halo pixels of the tile): pixel_terms (the same function pixel_pass1 / pixel_pass2
//     inline: bit-identical channel values), layer-norm statistics applied, the channels left in LDS as fp32 planes - one plane of 12 channels
//     per slot, one for the 6 channels the slots share; columns de-interleaved by parity as in kernels_refws.hip, 48-byte pixels (conflict-free
//     16-byte accesses); plane maxima by ds_max_u32.  In training the tile's own pixels also go to HBM in pixel_pass2's layout (the backward
//     reads them).  Loads of a unit are issued one unit ahead at clamped addresses;
//   * consumers (waves 0-3): first convert a quarter of the plane pixels each to fp16 hi | lo IN PLACE (scale = power of two of the plane
//     max), then, behind a barrier, each wave owns 16 output channels with both weight slices (12- and 8-channel part, 9 taps, hi + lo) in 72
//     VGPRs: v_mfma_f32_16x16x16_f16, the shared plane once per tile, then 54 MFMAs per slot; out = ELU(bias + shared + per-slot), 64 bytes
//     per pixel and wave.
// Two LDS-only barriers per unit (s_waitcnt lgkmcnt(0); s_barrier - not __syncthreads(), which also drains stores and prefetches).
// DESIGN.md 4.7 has the measurements and the forms tried on the way.
#include "common.h"
#include "pixel_terms.h"
#include "pack_bodies.h"
#include <utility>
#include <vector>
#include <cstdio>

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

#ifdef IODINE_TILE_PROF
__device__ unsigned g_l0_prof[2 * TP_MAXBLK * 8];            // [role][block][phase]: s_memtime ticks of consumer wave 0 / producer wave 4
#endif

constexpr int L0_HR = 5, L0_HC = 33, L0_NPX = L0_HR * L0_HC;      // halo of a 2 x 16 output tile: fine pixels
constexpr int L0_PXB = 48;                                         // bytes per staged pixel: 12 fp32 channels
constexpr int L0_PLB = L0_NPX * L0_PXB;                            // one plane: 7920 bytes

// w [O][CINW][9] -> [O / 16][9 taps][hi / lo][64 lanes] x 4 fp16: the A operand of v_mfma_f32_16x16x16_f16 (lane l: row = cout 16 g + l % 16,
// k = cin 4 (l / 16) .. + 3, zero past CINW), pre-scaled by meta[0]
__global__ void l0_pack_weights_kernel(const float* __restrict__ w, int O, int CINW, const float* __restrict__ meta, _Float16* __restrict__ dst)
{
    const float scale = meta[0];
    const int total = (int)pack_l0_total(O);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
        dst[idx] = pack_l0_element(w, CINW, scale, (size_t)idx);             // (pack_bodies.h: shared with the batched form)
}

__global__ __launch_bounds__(1024) void l0_weight_scale_kernel(const float* __restrict__ w, int n, float* __restrict__ meta)
{
    __shared__ float s_red[16];
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

IOD_DEVINL float l0_fresh_scale(float mx)
{
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
    int se = 12 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return __uint_as_float((unsigned)(127 + se) << 23);
}

// Block barrier that orders LDS traffic only: __syncthreads() is a full fence, i.e. s_waitcnt vmcnt(0) as well - on gfx9 that also drains the
// wave's global STORES (the consumers' output) and the producers' prefetch loads twice per unit.
IOD_DEVINL void l0_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 12 values * scale -> 12 fp16 hi | 12 fp16 lo at `dst` (48 bytes)
IOD_DEVINL void l0_store12(unsigned char* dst, const float (&v)[12], float scale)
{
    unsigned hi[6], lo[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) hi[q] = pack_hi_lo(v[2 * q] * scale, v[2 * q + 1] * scale, lo[q]);
    *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(dst + 16) = make_uint4(hi[4], hi[5], lo[0], lo[1]);
    *reinterpret_cast<uint4*>(dst + 32) = make_uint4(lo[2], lo[3], lo[4], lo[5]);
}

template <int K, bool ALLCH>                                 // ALLCH: every encoding channel present (chmask = 0x1ffff) - no per-channel selects
__global__ __launch_bounds__(448)
void refine_l0_fused_kernel(const float4* __restrict__ x4, const float4* __restrict__ dec, const float* __restrict__ lnstat,
                            const float* __restrict__ lin, const uint2* __restrict__ wk, const float* __restrict__ wkmeta,
                            const uint2* __restrict__ ws, const float* __restrict__ wsmeta, const float* __restrict__ bias,
                            float* __restrict__ out, float* __restrict__ enck, float* __restrict__ encs, int S, int nunits,
                            float inv2s2, float invs2, float lconst, unsigned chmask)
{
    constexpr int C = 64, NPL = K + 1;                        // planes: K per-slot + 1 shared
    constexpr int BUFB = NPL * L0_PLB;                        // one buffer of planes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_l0[];
    unsigned char* s_pl = smem_l0;                            // [2][NPL][L0_PLB]
    float* s_zero = reinterpret_cast<float*>(smem_l0 + 2 * BUFB);             // 16 bytes of zeros (pad channels of the fragments)
    unsigned* s_max = reinterpret_cast<unsigned*>(s_zero + 4);                // [3][NPL][16]: plane maxima (bit patterns of |v|: ordered as unsigned), ds_max_u32
                                                                              // from the producer lanes (lane % 16: 4-way conflicts); three buffers:
                                                                              // written for unit it, read for unit it - 1, zeroed for unit it + 1
    float* s_bias = reinterpret_cast<float*>(s_max + 3 * NPL * 16);           // [C]
    float* s_sc = s_bias + C;                                                 // [4 consumer waves][16]: plane scales of the unit being consumed

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Sc = S >> 1, P = S * S;
    const int tiles_x = Sc >> 4, tiles_y = Sc >> 1, tiles = tiles_x * tiles_y;
    const int n_my = (nunits - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int i = tid; i < 4 + 3 * NPL * 16; i += 448) s_zero[i] = 0.f;
    if (tid < C) s_bias[tid] = bias[tid];
    __syncthreads();

    if (wv < 4) {
        // =========================== consumers: wave = 16 output channels, weights in registers ===========================
        const int cg = wv, lpx = lane & 15, lkb = lane >> 4;
        f16x4 wkh[9], wkl[9], wsh[9], wsl[9];
        {
            const uint2* pk = wk + (size_t)cg * 9 * 2 * 64 + lane;
            const uint2* ps = ws + (size_t)cg * 9 * 2 * 64 + lane;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint2 a = pk[(t * 2 + 0) * 64], b = pk[(t * 2 + 1) * 64], c = ps[(t * 2 + 0) * 64], d = ps[(t * 2 + 1) * 64];
                __builtin_memcpy(&wkh[t], &a, 8); __builtin_memcpy(&wkl[t], &b, 8);
                __builtin_memcpy(&wsh[t], &c, 8); __builtin_memcpy(&wsl[t], &d, 8);
            }
        }
        const float inv_wk = wkmeta[1], inv_ws = wsmeta[1];
        const float4 bq = *reinterpret_cast<const float4*>(s_bias + 16 * cg + 4 * lkb);
        const unsigned zoff = (unsigned)(2 * BUFB);
        // The producers leave fp32 planes (the scale needs the maxima of all three producer waves).  Step 1 of a unit: the four consumer waves
        // turn them into fp16 hi | lo IN PLACE (48-byte pixel: 12 fp32 -> 12 hi + 12 lo fp16), a quarter of the plane pixels each - every wave
        // converting every fragment it uses (first form) was four times the work, and VALU issue slots are what this kernel runs out of (one
        // VALU instruction per 4 cycles and SIMD, shared by the producer and the consumer wave of that SIMD).  Step 2, behind a barrier:
        // fragment of (plane, halo row r, tap column kx) = 4 channels (k quarter lkb) of pixel column lpx as two ds_read_b64; quarters >= nq
        // read the zero slot.  Rows in the order 0 3 1 4 2: consecutive MFMAs alternate between the two accumulators.
        float* my_sc = s_sc + wv * 16;
        auto plane_max_scale = [&](const unsigned* mxp, int k) {
            const uint4* q = reinterpret_cast<const uint4*>(mxp + k * 16);
            const uint4 a = q[0], b2 = q[1], c = q[2], d = q[3];
            const unsigned m = max(max(max(max(a.x, a.y), max(a.z, a.w)), max(max(b2.x, b2.y), max(b2.z, b2.w))),
                                   max(max(max(c.x, c.y), max(c.z, c.w)), max(max(d.x, d.y), max(d.z, d.w))));
            return l0_fresh_scale(__uint_as_float(m));
        };
        auto convert_share = [&](unsigned bbase, const unsigned* mxp) {
            // lanes 0 .. NPL - 1: the scale of plane `lane` -> this wave's table (wave-private: LDS operations of a wave complete in order)
            my_sc[min(lane, 15)] = plane_max_scale(mxp, min(lane, NPL - 1));
            constexpr int TOT = NPL * L0_NPX, NIT = (TOT + 255) / 256;
#pragma unroll
            for (int q = 0; q < NIT; ++q) {
                const int e = wv * 64 + lane + 256 * q;
                if (e < TOT) {
                    const int plane = e / L0_NPX, pos = e - plane * L0_NPX;
                    const float sc = my_sc[plane];
                    float4* p = reinterpret_cast<float4*>(smem_l0 + bbase + (unsigned)(plane * L0_PLB + pos * L0_PXB));
                    const float4 v0 = p[0], v1 = p[1], v2 = p[2];
                    unsigned h[6], l[6];
                    h[0] = pack_hi_lo(v0.x * sc, v0.y * sc, l[0]); h[1] = pack_hi_lo(v0.z * sc, v0.w * sc, l[1]);
                    h[2] = pack_hi_lo(v1.x * sc, v1.y * sc, l[2]); h[3] = pack_hi_lo(v1.z * sc, v1.w * sc, l[3]);
                    h[4] = pack_hi_lo(v2.x * sc, v2.y * sc, l[4]); h[5] = pack_hi_lo(v2.z * sc, v2.w * sc, l[5]);
                    uint4* o = reinterpret_cast<uint4*>(p);
                    o[0] = make_uint4(h[0], h[1], h[2], h[3]); o[1] = make_uint4(h[4], h[5], l[0], l[1]); o[2] = make_uint4(l[2], l[3], l[4], l[5]);
                }
            }
        };
        auto conv_plane = [&](unsigned pbase, int nq, const f16x4 (&wh)[9], const f16x4 (&wl)[9], f32x4 (&acc)[2]) {
            const unsigned base = lkb < nq ? pbase + (unsigned)(lpx * L0_PXB + lkb * 8) : zoff;
            const unsigned step = lkb < nq ? (unsigned)L0_PXB : 0u;
            const unsigned lo_off = lkb < nq ? 24u : 0u;
            uint2 fh[15], fl[15];
#pragma unroll
            for (int r = 0; r < L0_HR; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int pos = r * L0_HC + (kx == 1 ? 17 : (kx == 2 ? 1 : 0));
                    const unsigned a = base + step * (unsigned)pos;
                    fh[r * 3 + kx] = *reinterpret_cast<const uint2*>(smem_l0 + a);
                    fl[r * 3 + kx] = *reinterpret_cast<const uint2*>(smem_l0 + a + lo_off);
                }
            acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
            for (int ri = 0; ri < L0_HR; ++ri) {
                const int r = ri == 0 ? 0 : (ri == 1 ? 3 : (ri == 2 ? 1 : (ri == 3 ? 4 : 2)));
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    f16x4 h4, l4;
                    __builtin_memcpy(&h4, &fh[r * 3 + kx], 8); __builtin_memcpy(&l4, &fl[r * 3 + kx], 8);
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        const int ky = r - 2 * y;
#ifdef L0_DBG_NOCONS
                        if (ky == 0 && kx == 0) acc[y] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[0], h4, acc[y], 0, 0, 0);
                        else if (ky >= 0 && ky <= 2) acc[y][0] += (float)l4[0];
                        continue;
#endif
                        if (ky >= 0 && ky <= 2) {
                            acc[y] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[ky * 3 + kx], l4, acc[y], 0, 0, 0);
                            acc[y] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl[ky * 3 + kx], h4, acc[y], 0, 0, 0);
                            acc[y] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[ky * 3 + kx], h4, acc[y], 0, 0, 0);
                        }
                    }
                }
            }
        };
        f32x4 sh[2];
        TP_DECL;
        for (int it = 0; it <= n_my; ++it) {
            TP_STAMP(0);
            const int j = it - 1, buf = j & 1;
            const unsigned bbase = (unsigned)(buf * BUFB);
            if (j >= 0) convert_share(bbase, s_max + (j % 3) * NPL * 16);
            TP_STAMP(1);
            l0_lds_barrier();                                 // every plane of the unit is fp16 now (the producers pass through)
            TP_STAMP(2);
            if (j >= 0) {
                const int u = (int)blockIdx.x + j * (int)gridDim.x;
                const int b = u / tiles, t = u % tiles, ty = t / tiles_x, tx = t % tiles_x;
                const int X = 16 * tx + lpx;
                {
                    f32x4 acc[2];
                    conv_plane(bbase + (unsigned)(K * L0_PLB), 2, wsh, wsl, acc);
                    const float inv = inv_ws / my_sc[K];
                    sh[0] = acc[0] * inv + f32x4{bq.x, bq.y, bq.z, bq.w};
                    sh[1] = acc[1] * inv + f32x4{bq.x, bq.y, bq.z, bq.w};
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    f32x4 acc[2];
                    conv_plane(bbase + (unsigned)(k * L0_PLB), 3, wkh, wkl, acc);
                    const float inv = inv_wk / my_sc[k];
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        const int Y = 2 * ty + y;
                        const f32x4 v = acc[y] * inv + sh[y];
                        *reinterpret_cast<float4*>(out + ((((size_t)b * K + k) * Sc + Y) * Sc + X) * C + 16 * cg + 4 * lkb) =
                            make_float4(elu1_fast(v.x), elu1_fast(v.y), elu1_fast(v.z), elu1_fast(v.w));
                    }
                }
            }
            TP_STAMP(3);
            l0_lds_barrier();
            TP_STAMP(4);
        }
#ifdef IODINE_TILE_PROF
        if (tid == 0 && blockIdx.x < TP_MAXBLK) for (int i_ = 0; i_ < 8; ++i_) g_l0_prof[blockIdx.x * 8 + i_] = tp_acc[i_];
#endif
        return;
    }
    // =========================== producers: thread = halo pixel (row pr, column ph), all slots ===========================
    auto on = [&](int c, float v) { return ALLCH || ((chmask >> c) & 1u) ? v : 0.f; };
    const int q = tid - 256, pw = wv - 4;
    const bool pixel_thread = q < L0_NPX;
    const int pr = q / L0_HC, ph = q % L0_HC;
    const int ppos = pr * L0_HC + ((ph & 1) ? 17 + (ph >> 1) : (ph >> 1));     // columns de-interleaved by parity
    float4 nxv = make_float4(0.f, 0.f, 0.f, 0.f), ndv[K];
    float ncxy[2] = {0.f, 0.f};                               // (every global load of a unit is issued one unit ahead: vmcnt is in-order, a load issued
                                                              //  behind the prefetch and used at once would wait for the whole prefetch)
#pragma unroll
    for (int k = 0; k < K; ++k) ndv[k] = nxv;
    auto unit_pixel = [&](int j, int& b, int& fy, int& fx) {
        const int u = (int)blockIdx.x + j * (int)gridDim.x;
        b = u / tiles;
        const int t = u % tiles, ty = t / tiles_x, tx = t % tiles_x;
        fy = 4 * ty - 1 + pr; fx = 32 * tx - 1 + ph;
        return pixel_thread && j < n_my && (unsigned)fy < (unsigned)S && (unsigned)fx < (unsigned)S;
    };
    // every load is issued unconditionally at a clamped (always valid) address - a branch around the loads makes hipcc wait for them at the
    // join; what lies outside the image (or belongs to no unit) is zeroed where it is used
    auto prefetch = [&](int j) {
        int b, fy, fx;
        (void)unit_pixel(j, b, fy, fx);
        const int bs = j < n_my ? b : 0;
        const int cy = min(max(fy, 0), S - 1), cx = min(max(fx, 0), S - 1);
        const size_t p = (size_t)cy * S + cx;
        nxv = x4[(size_t)bs * P + p];
#pragma unroll
        for (int k = 0; k < K; ++k) ndv[k] = dec[((size_t)bs * K + k) * P + p];
        ncxy[0] = lin[cx]; ncxy[1] = lin[cy];
    };
    prefetch(0);
    TP_DECL;
    for (int it = 0; it <= n_my; ++it) {
        TP_STAMP(0);
        const int buf = it & 1;
        int b, fy, fx;
        const bool inside = unit_pixel(it, b, fy, fx);
        float lnv[K][8];                                      // layer-norm statistics of the image: wave-uniform (scalar loads), fetched up front
        {
            const float* lnb = lnstat + (size_t)(it < n_my ? b : 0) * K * 8;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int c = 0; c < 8; ++c) lnv[k][c] = lnb[k * 8 + c];
        }
        l0_lds_barrier();                                     // the consumers' in-place conversion of the previous unit (they wait for every wave)
        PixelTerms<K> tm;
        const float4 xv = nxv;
        float psum = 1.f;
        const float cxy[2] = {ncxy[0], ncxy[1]};
        // channel values of slot k / of the shared plane from the pixel terms (evaluated twice: for the plane maxima, then for the stores)
        auto slot_vals = [&](int k, float (&v)[12]) {
            const float* ln = lnv[k];
            v[0] = on(3, tm.mu[k][0]); v[1] = on(4, tm.mu[k][1]); v[2] = on(5, tm.mu[k][2]);
            v[3] = on(6, tm.m[k]); v[4] = on(7, tm.logit[k]); v[5] = on(8, tm.pk[k] / psum);
            v[6] = on(9, (tm.g1[k][0] - ln[0]) * ln[1]); v[7] = on(10, (tm.g1[k][1] - ln[0]) * ln[1]);
            v[8] = on(11, (tm.g1[k][2] - ln[0]) * ln[1]); v[9] = on(12, (tm.g2[k] - ln[2]) * ln[3]);
            v[10] = on(14, (tm.loo[k] - ln[4]) * ln[5]); v[11] = 0.f;
        };
        auto shared_vals = [&](float (&v)[12]) {
            v[0] = on(0, xv.x); v[1] = on(1, xv.y); v[2] = on(2, xv.z);
            v[3] = on(13, (tm.like - lnv[0][6]) * lnv[0][7]);       // LN statistics of slot 0: the same for every slot
            v[4] = on(15, cxy[0]); v[5] = on(16, cxy[1]);
#pragma unroll
            for (int c = 6; c < 12; ++c) v[c] = 0.f;
        };
        if (it < n_my) {
            float4 dv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) dv[k] = ndv[k];
            prefetch(it + 1);
            {
                {
#ifdef L0_DBG_NOPROD
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        tm.mu[k][0] = dv[k].x; tm.mu[k][1] = dv[k].y; tm.mu[k][2] = dv[k].z; tm.m[k] = dv[k].w; tm.logit[k] = dv[k].w; tm.pk[k] = 1.f;
                        tm.g1[k][0] = dv[k].x; tm.g1[k][1] = dv[k].y; tm.g1[k][2] = dv[k].z; tm.g2[k] = dv[k].w; tm.loo[k] = dv[k].x;
                    }
                    tm.like = xv.x;
#else
                    pixel_terms_core<K>(xv, dv, inv2s2, invs2, lconst, tm);
#endif
                    psum = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) psum += tm.pk[k];
                }
                // the planes hold the fp32 values (the consumers scale and split them: the scale needs the maxima of all producer waves)
                const bool wr = inside && enck && pr >= 1 && ph >= 1;     // training: the tile's own 4 x 32 pixels also go to HBM (pixel_pass2's layout)
                const size_t p = (size_t)fy * S + fx;
                unsigned* mxw = s_max + ((it % 3) * NPL) * 16 + (lane & 15);
                if (q < NPL * 16) s_max[(((it + 1) % 3) * NPL) * 16 + q] = 0u;
                if (pixel_thread) {
                    if (inside) {
#pragma unroll
                        for (int k = 0; k <= K; ++k) {
                            float v[12];
                            if (k < K) slot_vals(k, v);
                            else shared_vals(v);
                            float m = 0.f;
#pragma unroll
                            for (int c = 0; c < 11; ++c) m = fmaxf(m, fabsf(v[c]));
                            atomicMax(mxw + k * 16, __float_as_uint(m));
                            float4* d = reinterpret_cast<float4*>(s_pl + buf * BUFB + k * L0_PLB + ppos * L0_PXB);
                            const float4 q0 = make_float4(v[0], v[1], v[2], v[3]), q1 = make_float4(v[4], v[5], v[6], v[7]), q2 = make_float4(v[8], v[9], v[10], v[11]);
                            d[0] = q0; d[1] = q1; d[2] = q2;
                            if (wr) {
                                if (k < K) {
                                    float4* o = reinterpret_cast<float4*>(enck + (((size_t)b * K + k) * P + p) * 12);
                                    o[0] = q0; o[1] = q1; o[2] = q2;
                                } else {
                                    float4* os = reinterpret_cast<float4*>(encs + ((size_t)b * P + p) * 8);
                                    os[0] = q0; os[1] = q1;
                                }
                            }
                        }
                    } else {                                  // zero padding of the conv (border tiles only)
                        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int k = 0; k <= K; ++k) {
                            float4* d = reinterpret_cast<float4*>(s_pl + buf * BUFB + k * L0_PLB + ppos * L0_PXB);
                            d[0] = z; d[1] = z; d[2] = z;
                        }
                    }
                }
            }
        }
        TP_STAMP(1);
        l0_lds_barrier();
        TP_STAMP(2);
    }
#ifdef IODINE_TILE_PROF
    if (tid == 256 && blockIdx.x < TP_MAXBLK) for (int i_ = 0; i_ < 8; ++i_) g_l0_prof[(TP_MAXBLK + blockIdx.x) * 8 + i_] = tp_acc[i_];
#endif
}

template <int K>
hipError_t l0_launch(hipStream_t st, const float* x4, const float* dec, const float* lnstat, const float* lin, const void* wk, const float* wkmeta,
                     const void* ws, const float* wsmeta, const float* bias, float* out, float* enck, float* encs, int B, int S, float sigma,
                     unsigned chmask)
{
    constexpr size_t lds = (size_t)2 * (K + 1) * L0_PLB + 16 + (size_t)3 * (K + 1) * 64 + 64 * 4 + 64 * 4 + 64;
    static std::atomic<unsigned> attr_devs{0}, attr_devs2{0};
    if (hipError_t e = iod_set_max_lds((const void*)refine_l0_fused_kernel<K, true>, (int)lds, attr_devs); e != hipSuccess) return e;
    if (hipError_t e = iod_set_max_lds((const void*)refine_l0_fused_kernel<K, false>, (int)lds, attr_devs2); e != hipSuccess) return e;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    const int Sc = S / 2, nunits = B * (Sc / 16) * (Sc / 2);
    const int blocks = std::min(nunits, n_cu);
    const float inv2s2 = 1.f / (2.f * sigma * sigma), invs2 = 1.f / (sigma * sigma);
    const float lconst = (float)(-log((double)sigma) - 0.5 * log(2.0 * M_PI));
    if ((chmask & 0x1ffffu) == 0x1ffffu)
        hipLaunchKernelGGL((refine_l0_fused_kernel<K, true>), dim3(blocks), dim3(448), lds, st, (const float4*)x4, (const float4*)dec, lnstat, lin,
                           (const uint2*)wk, wkmeta, (const uint2*)ws, wsmeta, bias, out, enck, encs, S, nunits, inv2s2, invs2, lconst, chmask);
    else
        hipLaunchKernelGGL((refine_l0_fused_kernel<K, false>), dim3(blocks), dim3(448), lds, st, (const float4*)x4, (const float4*)dec, lnstat, lin,
                           (const uint2*)wk, wkmeta, (const uint2*)ws, wsmeta, bias, out, enck, encs, S, nunits, inv2s2, invs2, lconst, chmask);
#ifdef IODINE_TILE_PROF
    {
        const int nb = std::min(blocks, TP_MAXBLK);
        std::vector<unsigned> hp((size_t)2 * TP_MAXBLK * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_l0_prof), hp.size() * sizeof(unsigned));
        static const char* pn[5] = {"loop", "convert / work", "barrier", "mfma / -", "barrier"};
        for (int role = 0; role < 2; ++role) {
            double sum[8] = {0}, tot = 0;
            for (int b2 = 0; b2 < nb; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[((size_t)role * TP_MAXBLK + b2) * 8 + i];
            for (int i = 0; i < 8; ++i) tot += sum[i] / nb;
            fprintf(stderr, "[refl0 prof] %s ticks per block (%d units), total %.0f:", role ? "producer" : "consumer", (nunits + blocks - 1) / blocks, tot);
            for (int i = 0; i < 5; ++i) fprintf(stderr, " %s %.0f |", pn[i], sum[i] / nb);
            fprintf(stderr, "\n");
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace

size_t refine_l0_wpk_bytes(int O) { return (size_t)(O / 16) * 9 * 2 * 64 * 8; }

// w [O][cinw][9] (cinw = 12: per-slot part, 8: per-image part; internal channel order of ref_split_weights) -> register layout + {scale, 1 / scale}
hipError_t launch_refine_l0_pack(hipStream_t st, const float* w, int O, int cinw, float* meta, void* dst)
{
    hipLaunchKernelGGL(l0_weight_scale_kernel, dim3(1), dim3(1024), 0, st, w, O * cinw * 9, meta);
    const int total = (O / 16) * 9 * 2 * 64 * 4;
    hipLaunchKernelGGL(l0_pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, O, cinw, meta, (_Float16*)dst);
    return hipGetLastError();
}

bool refine_l0_fused_ok(int S, int c, int K) { return c == 64 && S >= 32 && S % 32 == 0 && K >= 1 && K <= 9; }

// Encoding + first refinement layer for all K slots of B images: out [B K][S/2][S/2][64] = ELU(conv_s2(encoding) + bias).
// enck / encs (both or neither): also write the split encoding (pixel_pass2's layout) - training.
hipError_t launch_refine_l0_fused(hipStream_t st, const float* x4, const float* dec, const float* lnstat, const float* lin, const void* wk,
                                  const float* wkmeta, const void* ws, const float* wsmeta, const float* bias, float* out, float* enck,
                                  float* encs, int B, int K, int S, int c, float sigma, unsigned chmask)
{
    IOD_XSKIP(512);
    if (!refine_l0_fused_ok(S, c, K) || (enck == nullptr) != (encs == nullptr)) return hipErrorInvalidValue;
    switch (K) {
#define L0_CASE(KK) case KK: return l0_launch<KK>(st, x4, dec, lnstat, lin, wk, wkmeta, ws, wsmeta, bias, out, enck, encs, B, S, sigma, chmask);
        L0_CASE(1) L0_CASE(2) L0_CASE(3) L0_CASE(4) L0_CASE(5) L0_CASE(6) L0_CASE(7) L0_CASE(8) L0_CASE(9)
#undef L0_CASE
        default: return hipErrorInvalidValue;
    }
}
