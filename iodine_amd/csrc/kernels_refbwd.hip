// Backward of the first two layers of the refinement conv stack in ONE pass (training, round 4).  gfx950 only.
//
// Reference: RefinementNetwork.mlc (lib/modeling/iodine.py:459,480: REF.CONV_LAYERS x [conv k3 s2 p1 + ELU]) under the outer
// loss.backward() (lib/engine/train.py:63).  Layer 0's input is detached (iodine.py:343), so its backward is a weight gradient
// only, and the gradient wrt its pre-activation
//     dpre0 = convT_s2(W1, dpre1) * ELU'(act0)                                   (the data gradient of layer 1)
// has exactly ONE consumer:  dW0[tap][ci][co] = sum_px enc[2y+ky-1][2x+kx-1][ci] * dpre0[y][x][co],  db0 = sum_px dpre0.
// Until round 3 these were two launches with the 1.17 GB tensor dpre0 (T*N slot-images x 64 x 64 x 64 ch at cfg3) written by the
// first and read back by the second: 0.72 + 0.74 ms, 1.42x / 1.28x their algorithmic bytes.  Here dpre0 never exists in memory:
//
//   512 threads = one persistent block per CU, tile = 4 x 16 pixels of the 64 x 64 grid, warp-specialised:
//   * waves 4-7 (PRODUCERS) each own 16 channels of dpre0 and keep the matching slice of W1 (all 9 taps x 64 input channels, fp16
//     hi + lo: 144 VGPRs, the register layout of the weight-stationary decoder conv) for the life of the block.  Per tile they
//     build the B fragments of v_mfma_f32_16x16x32_f16 straight from global memory (dpre1 window (oy, ox) of the 3 x 9 coarse
//     neighbourhood: lane = (pixel, 8 consecutive channels), split into fp16 hi / lo in registers), issue the 54 MFMAs of the four
//     parity classes (1 + 2 + 2 + 4 taps: a stride-2 transposed conv has no work on structural zeros), multiply by ELU'(act0), and
//     stage the 20-channel encoding halo (9 x 33 pixels of the 128 x 128 input: 12 per-slot + 8 per-image channels) - both as the
//     transposed fp16 planes the weight-gradient MFMAs read (K = pixels), scaled by a block-wide power of two;
//   * waves 0-3 (CONSUMERS) hold the accumulators of dW0 in GEMM form (rows (kx, ky, ci): 6 tiles of 32 x 32, 96 VGPRs), read fragments,
//     issue 36 MFMAs per tile, and stage the encoding halo + the coarse gradient neighbourhood for the producers.
//   One s_barrier per tile; planes double-buffered; the raw values of tile t+2 are produced (and their max |.| published) while
//   tile t is consumed, packed with the block-wide scale one iteration later.
// Traffic per launch at cfg3: dpre1 0.29 GB + act0 1.17 GB + encoding 0.97 GB read, 19 MB of partial tiles written.
#include "common.h"
#include <utility>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int RB_TH = 4;                         // tile rows (64 x 64 grid); 16 columns
constexpr int RB_HR = 2 * RB_TH + 1;             // input halo rows of the 128 x 128 grid
constexpr int RB_APL = RB_HR * 20;               // dwords per encoding channel plane (2 px per dword; APL / 4 odd: conflict-free b128)
constexpr int RB_DPL = RB_TH * 8 + 4;            // dwords per gradient channel plane
constexpr int RB_CI = 32, RB_CIR = 20, RB_CO = 64;
static_assert((RB_APL / 4) % 2 == 1 && (RB_DPL / 4) % 2 == 1, "plane strides");
constexpr int RB_A_DW = 2 * RB_CI * RB_APL;      // one buffer of encoding planes [hi/lo][32][APL]
constexpr int RB_D_DW = 2 * RB_CO * RB_DPL;      // one buffer of gradient planes [hi/lo][64][DPL]
constexpr int RB_RPX = 68;                       // dwords per staged coarse-gradient pixel (64 + 4 pad: spreads the fragment reads over the banks)
constexpr int RB_R_DW = 27 * RB_RPX;             // one buffer of raw fp32 dpre1 values: the 3 x 9 coarse neighbourhood of a tile
constexpr size_t RB_LDS = (size_t)(2 * RB_A_DW + 2 * RB_D_DW + 2 * RB_R_DW) * 4 + 192;

IOD_DEVINL float rb_fresh_scale(float mx)
{
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
    int se = 12 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return __uint_as_float((unsigned)(127 + se) << 23);
}

struct RbFrag { f16x8 h, l; };

#ifdef IODINE_TILE_PROF
__device__ unsigned g_rb_prof[2 * TP_MAXBLK * 8];           // [role][block][phase]: s_memtime ticks of consumer wave 0 / producer wave 4
#endif

// 8 fp32 values (two float4) * scale -> fp16 hi / lo fragments
IOD_DEVINL RbFrag rb_split8(const float4 a, const float4 b, float scale)
{
    const float v[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale, b.x * scale, b.y * scale, b.z * scale, b.w * scale};
    unsigned hi[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hi[q] = pack_hi_lo(v[2 * q], v[2 * q + 1], lo[q]);
    RbFrag f;
    __builtin_memcpy(&f.h, hi, 16);
    __builtin_memcpy(&f.l, lo, 16);
    return f;
}

typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));

// raw buffer load of 16 bytes: an offset with bit 31 set is out of range of the 2 GB descriptor and returns zeros in hardware (no branch)
IOD_DEVINL float4 rb_bload(__amdgpu_buffer_rsrc_t r, int voff)
{
    const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
IOD_DEVINL __amdgpu_buffer_rsrc_t rb_rsrc(const float* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, 0x7fffffff, 0x00020000); }

__global__ __launch_bounds__(512, 1)
void refine_bwd01_kernel(const float* __restrict__ rd1, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                         const float* __restrict__ act0, const float* __restrict__ enck, const float* __restrict__ encs,
                         float* __restrict__ part, float* __restrict__ part_b, int Sm, int lgSm, int ntiles, unsigned kdiv_magic)
{
    constexpr int C = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned smem_rb[];
    unsigned* s_a = smem_rb;                                  // [2 buf][2 term][32][APL]
    unsigned* s_d = smem_rb + 2 * RB_A_DW;                    // [2 buf][2 term][64][DPL]
    float* s_rd = reinterpret_cast<float*>(s_d + 2 * RB_D_DW);     // [2 buf][27 px][RPX] raw dpre1 neighbourhood (fp32)
    float* s_max = s_rd + 2 * RB_R_DW;                        // [2 parity][12]: 0..3 encoding max (consumer waves), 4..7 gradient max (producer
                                                              // waves), 8..11 max of the coarse neighbourhood (consumer waves)
    float* s_sc = s_max + 24;                                 // [2 buf][2]: scales the planes of a buffer were packed with

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int Sc = Sm >> 1, Sf = Sm << 1;
    // (Sm is a power of two - the launcher checks - so tile coordinates are shifts: 16 columns x 4 rows per tile)
    const int lg_tx = lgSm - 4, lg_ty = lgSm - 2, tiles_x = 1 << lg_tx, tiles_y = 1 << lg_ty;

    // persistent, XCD-aware schedule (block b runs on XCD b % 8, observed; speed only): every XCD walks one contiguous eighth of
    // the tile list, so that blocks sharing an L2 work on neighbouring tiles (their halos overlap)
    const int nblk = gridDim.x;
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = (nblk + 7) >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_begin = xcd * per_xcd, t_end = min(ntiles, t_begin + per_xcd);
    const int t0 = t_begin + bix;
    const int ntl = t0 < t_end ? (t_end - t0 + bpx - 1) / bpx : 0;
    auto tile_of = [&](int j, int& nimg, int& ty, int& tx) {           // block-uniform: scalar ALU
        const int t = t0 + j * bpx;
        tx = t & (tiles_x - 1);
        ty = (t >> lg_tx) & (tiles_y - 1);
        nimg = t >> (lg_tx + lg_ty);
    };

    // pad channel planes (20..31) stay zero for the whole kernel
    for (int i = tid; i < 2 * RB_A_DW; i += 512) s_a[i] = 0u;
    if (tid < 24) s_max[tid] = 0.f;

    // ---- The coarse neighbourhood (3 x 9 pixels x 64 channels of dpre1, fp32) of tile j is staged through LDS by the CONSUMERS: loaded
    // into registers during iteration j - 4 (latency hidden by that iteration's MFMAs), stored - and its max |.| published - during
    // iteration j - 3, read by the producers in iteration j - 2.  (Fragments loaded by the producers straight from global memory left
    // two exposed HBM round trips per tile: 5.6 us per tile.)  Everything a thread needs per unit is a constant of the thread: byte
    // offset relative to the tile's origin, LDS slot, border flags - the per-tile part is scalar arithmetic + a buffer descriptor.
    float4 rr1[2];                                            // 432 float4 per tile over the 256 consumer threads
    int r_voff[2], r_lds[2], r_flg[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = (tid & 255) + k * 256;
        const int px = u >> 4, seg = u & 15, pyy = px / 9, pxx = px % 9;
        r_voff[k] = ((pyy * Sc + pxx) * C + seg * 4) * 4;
        r_lds[k] = px * RB_RPX + seg * 4;
        r_flg[k] = (u < 27 * 16 ? 4 : 0) | (pyy == 2 ? 1 : 0) | (pxx == 8 ? 2 : 0);
    }
    auto rd_load = [&](int j) {
        int nimg, ty, tx;
        tile_of(j, nimg, ty, tx);
        const __amdgpu_buffer_rsrc_t rs = rb_rsrc(rd1 + ((size_t)nimg * Sc * Sc + (size_t)(2 * ty) * Sc + 8 * tx) * C);
        const bool last_y = ty == tiles_y - 1, last_x = tx == tiles_x - 1;     // the +1 row / column of the neighbourhood is outside the image
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool inval = !(r_flg[k] & 4) || (last_y && (r_flg[k] & 1)) || (last_x && (r_flg[k] & 2));
            rr1[k] = rb_bload(rs, inval ? (int)0x80000000 : r_voff[k]);
        }
    };
    auto rd_store = [&](int j) {
        float* dst = s_rd + (j & 1) * RB_R_DW;
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (r_flg[k] & 4) *reinterpret_cast<float4*>(dst + r_lds[k]) = rr1[k];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(rr1[k].x), fabsf(rr1[k].y)), fmaxf(fabsf(rr1[k].z), fabsf(rr1[k].w))));
        }
        m = wave_max_f32(m);
        if (lane == 0) s_max[(j & 1) * 12 + 8 + wv] = m;
    };
    if (!producer)
        for (int j = 0; j < 2; ++j)
            if (j < ntl) { rd_load(j); rd_store(j); }                // tiles 0 and 1 up front
    __syncthreads();

    if (!producer) {
        // ================================= CONSUMERS: dW0 += enc^T . dpre0 ====================================================
        const int kh = lane >> 5, li = lane & 31;
        const int ni = wv & 1, ks = wv >> 1;                  // co 32-group, K split (tile rows 2 ks, 2 ks + 1)
        const int co = ni * 32 + li;
        // GEMM form of the weight gradient: rows (kx, ky, ci) instead of nine [32 ci x 32 co] tap tiles with 12 of 32 rows padding -
        // every kx owns 64 rows = 3 ky x 20 ci + 4 pad rows, i.e. two 32-row MFMA blocks with a block-uniform column shift:
        // 6 accumulator tiles (96 VGPRs) and 36 MFMAs per wave and tile instead of 9 / 144 / 54.  Lane li of row block rb holds
        // local row 32 (rb & 1) + li = 20 ky + ci; pad rows read the all-zero plane of channel 20.
        int offA[2];                                          // dword offset of this lane's (ci plane, ky row) for rb & 1 = 0, 1
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int local = 32 * q + li;
            const int ky = local / 20, cch = local % 20;
            offA[q] = local < 60 ? cch * RB_APL + ky * 20 : 20 * RB_APL;
        }
        f32x16 acc[6];
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float acc_prod = 1.f;
        // ---- encoding halo staging (the consumers' MFMA phase hides its loads; the producers' registers are full of weights).
        // Unit = (halo row, column slot, channel quad): two pixels of one x parity, four channels -> 4 packed dwords of 4 channel planes.
        // Unit slots k = 0, 1 read the per-slot tensor (3 quads x 153 positions = 459 units), k = 2, 3 the per-image tensor (2 x 153 = 306):
        // one buffer descriptor per k.  Byte offsets are relative to pixel (8 ty - 1, 32 tx - 4) of the tile's image.
        float4 ra[4][2];
        int e_voff[4], e_lds[4], e_flg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool img = k >= 2;
            const int nq = img ? 2 : 3, pst = img ? 8 : 12;
            const int idx = tid + 256 * (k & 1);
            const int q = idx % nq, tt = idx / nq;
            const int c4 = img ? 3 + q : q;
            const int slot = tt % 17, row = tt / 17;
            const int par = slot <= 8 ? 1 : 0;
            const int j = slot == 0 ? -1 : (slot <= 8 ? slot - 1 : slot - 9);
            e_voff[k] = ((row * Sf + 4 * j + 4 + par) * pst + q * 4) * 4;
            e_lds[k] = (c4 * 4) * RB_APL + row * 20 + slot + 3;
            e_flg[k] = (idx < 153 * nq ? 16 : 0) | (row == 0 ? 1 : 0) | (slot == 0 ? 2 : 0) | ((c4 & 3) << 2);
        }
        float sa = 1.f;
        TP_DECL;
        for (int it = -2; it < ntl; ++it) {
            TP_STAMP(0);                                         // [0] loop overhead
            if (it + 1 >= 0 && it + 1 < ntl) {
                // pack the raw values of tile it + 1 (in ra since the previous iteration) into plane buffer (it + 1) & 1
                const float* sm = s_max + ((it + 1) & 1) * 12;
                sa = tile_scale(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])), sa);
                const int buf = (it + 1) & 1;
                if (tid == 0) s_sc[buf * 2] = sa;
                unsigned* pa = s_a + buf * RB_A_DW;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (e_flg[k] & 16) {
                        const int rot = (e_flg[k] >> 2) & 3;
                        const float4 q0 = rot4(ra[k][0], rot), q1 = rot4(ra[k][1], rot);
                        const float x0[4] = {q0.x, q0.y, q0.z, q0.w}, x1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned lo;
                            const unsigned hi = pack_hi_lo(x0[e] * sa, x1[e] * sa, lo);
                            const int o = e_lds[k] + ((e + rot) & 3) * RB_APL;
                            pa[o] = hi;
                            pa[RB_CI * RB_APL + o] = lo;
                        }
                    }
                }
            }
            TP_STAMP(1);                                         // [1] encoding pack -> planes
            if (it + 3 >= 2 && it + 3 < ntl) rd_store(it + 3);      // loaded in the previous iteration; buffer (it + 3) & 1 was last read for tile it + 1
            if (it + 4 < ntl) rd_load(it + 4);
            if (it + 2 < ntl) {
                // encoding halo loads of tile it + 2 (consumed one iteration later; in flight under this iteration's MFMAs)
                int nimg, ty, tx;
                tile_of(it + 2, nimg, ty, tx);
                const long long org = (long long)(8 * ty - 1) * Sf + 32 * tx - 4;       // halo origin pixel (may lie before the image: masked)
                const int nim2 = kdiv_magic ? (int)__umulhi((unsigned)nimg, kdiv_magic) : nimg;   // nimg / K (the launcher validated the multiplier; 0: K = 1)
                const __amdgpu_buffer_rsrc_t rs_k = rb_rsrc(enck + ((long long)nimg * Sf * Sf + org) * 12);
                const __amdgpu_buffer_rsrc_t rs_s = rb_rsrc(encs + ((long long)nim2 * Sf * Sf + org) * 8);
                const bool top = ty == 0, left = tx == 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool inval = !(e_flg[k] & 16) || (top && (e_flg[k] & 1)) || (left && (e_flg[k] & 2));
                    const int vo = inval ? (int)0x80000000 : e_voff[k];
                    ra[k][0] = rb_bload(k < 2 ? rs_k : rs_s, vo);
                    ra[k][1] = rb_bload(k < 2 ? rs_k : rs_s, vo + (k < 2 ? 96 : 64));      // the pixel two columns on (same parity)
                }
            }
            TP_STAMP(2);                                         // [2] coarse-gradient store + load issue, encoding load issue
            if (it >= 0) {
                const int buf = it & 1;
                const float prod = s_sc[buf * 2] * s_sc[buf * 2 + 1];
                if (prod != acc_prod) {                          // block-uniform; exact (powers of two)
                    const float r = prod / acc_prod;
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp)
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc[tp][q] *= r;
                    acc_prod = prod;
                }
                const unsigned* sa_p = s_a + buf * RB_A_DW;
                const unsigned* sd_p = s_d + buf * RB_D_DW;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int r = ks * 2 + rr;
                    h16x8 bh, bl;
                    {
                        const uint4 vb_h = *reinterpret_cast<const uint4*>(sd_p + (0 * RB_CO + co) * RB_DPL + r * 8 + 4 * kh);
                        const uint4 vb_l = *reinterpret_cast<const uint4*>(sd_p + (1 * RB_CO + co) * RB_DPL + r * 8 + 4 * kh);
                        __builtin_memcpy(&bh, &vb_h, 16); __builtin_memcpy(&bl, &vb_l, 16);
                    }
#pragma unroll
                    for (int rb = 0; rb < 6; ++rb) {
                        const int kx = rb >> 1;                  // block-uniform column tap: 0 -> odd plane shifted by one pixel, 1 -> even plane, 2 -> odd plane
                        h16x8 A[2];
#pragma unroll
                        for (int term = 0; term < 2; ++term) {
                            const unsigned* pl = sa_p + term * RB_CI * RB_APL + offA[rb & 1] + (2 * r) * 20;
                            uint4 v;
                            if (kx == 1) v = *reinterpret_cast<const uint4*>(pl + 12 + 4 * kh);               // even plane X = 8 kh .. 8 kh + 7
                            else v = *reinterpret_cast<const uint4*>(pl + 4 + 4 * kh);                        // odd plane
                            if (kx == 0) {
                                const unsigned prev = pl[3 + 4 * kh] & 0xffff0000u;                            // odd X = 8 kh - 1 in the high half
                                uint4 m1;
                                m1.x = __builtin_amdgcn_alignbit(v.x, prev, 16);
                                m1.y = __builtin_amdgcn_alignbit(v.y, v.x, 16);
                                m1.z = __builtin_amdgcn_alignbit(v.z, v.y, 16);
                                m1.w = __builtin_amdgcn_alignbit(v.w, v.z, 16);
                                v = m1;
                            }
                            __builtin_memcpy(&A[term], &v, 16);
                        }
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], bh, acc[rb], 0, 0, 0);
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], bl, acc[rb], 0, 0, 0);
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], bh, acc[rb], 0, 0, 0);
                    }
                }
            }
            TP_STAMP(3);                                         // [3] fragment reads + MFMAs
            if (it + 2 < ntl) {
                float ma = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        ma = fmaxf(ma, fmaxf(fmaxf(fabsf(ra[k][q].x), fabsf(ra[k][q].y)), fmaxf(fabsf(ra[k][q].z), fabsf(ra[k][q].w))));
                ma = wave_max_f32(ma);
                if (lane == 0) s_max[((it + 2) & 1) * 12 + wv] = ma;
            }
            TP_STAMP(4);                                         // [4] wait for the encoding loads + max
            __syncthreads();
            TP_STAMP(5);                                         // [5] barrier
        }
#ifdef IODINE_TILE_PROF
        if (tid == 0 && blockIdx.x < TP_MAXBLK) for (int i_ = 0; i_ < 8; ++i_) g_rb_prof[blockIdx.x * 8 + i_] = tp_acc[i_];
#endif
        const float inv = 1.f / acc_prod;
        // partial tile [9 taps][20 ci][64 co] of this (block, K half)
        float* pw = part + (size_t)(blockIdx.x * 2 + ks) * 9 * RB_CIR * RB_CO;
#pragma unroll
        for (int rb = 0; rb < 6; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int local = 32 * (rb & 1) + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (local < 60) {
                    const int tap = (local / 20) * 3 + (rb >> 1), cch = local % 20;
                    pw[((size_t)tap * RB_CIR + cch) * RB_CO + ni * 32 + li] = acc[rb][r] * inv;
                }
            }
        return;
    }

    // ===================================== PRODUCERS ==========================================================================
    const int ptid = tid - 256;
    const int g = wv - 4;                                     // channel group of dpre0 this wave computes: 16 g .. 16 g + 15
    const int n = lane & 15, cq = lane >> 4;                  // MFMA column (pixel of a parity class) / k quarter resp. channel quad
    const int pi = n >> 3, pj = n & 7;                        // class pixel (row pair, column pair) of the tile

    // W1 slice -> registers: A operand of v_mfma_f32_16x16x32_f16, row = channel 16 g + n, k = 8 cq + e of chunk c
    // (launch_pack_conv_weights_ws with tflip 1 stores W1[k][row][8 - tap] at tap index `tap`)
    f16x8 wh[2][9], wl[2][9];
    {
        const uint4* wp = wpk + (size_t)g * 2 * 9 * 2 * 64 + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint4 h = wp[((c * 9 + (8 - t)) * 2 + 0) * 64], l = wp[((c * 9 + (8 - t)) * 2 + 1) * 64];
                __builtin_memcpy(&wh[c][t], &h, 16);
                __builtin_memcpy(&wl[c][t], &l, 16);
            }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(wh[c][t]), "+v"(wl[c][t]));
    }
    const float inv_w = wmeta[1];

    // per-thread constants: ELU' operand offset inside the tile (bytes), fragment position in the staged neighbourhood, plane slot
    const int a_voff = (((2 * pi) * Sm + 2 * pj) * C + 16 * g + 4 * cq) * 4;
    const int f_off = (pi * 9 + pj) * RB_RPX + 8 * cq;
    const int d_off = (16 * g + 4 * cq) * RB_DPL + (2 * pi) * 8 + pj;
    float4 ax[2][2];                                          // ELU' operand (act0, this lane's 4 channels at its pixel of every parity class) of the NEXT tile to compute
    auto ax_load = [&](int j) {
        int nimg, ty, tx;
        tile_of(j, nimg, ty, tx);
        const __amdgpu_buffer_rsrc_t rs = rb_rsrc(act0 + ((size_t)nimg * Sm * Sm + (size_t)(ty * RB_TH) * Sm + tx * 16) * C);
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const int vo = a_voff + py * Sm * C * 4;
            ax[py][0] = rb_bload(rs, vo);
            ax[py][1] = rb_bload(rs, vo + C * 4);
        }
    };
    f32x4 dv[2][2];                                           // raw gradient values of the tile in flight: [py][px], 4 channels of pixel n
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float sd = 1.f;

    // raw gradient values of tile j -> dv; publishes this wave's max |.|
    auto fetch_compute = [&](int j) {
        // ---- data gradient of layer 1 for this tile: four parity classes, two chunks of 32 input channels ----
        const float* rdl = s_rd + (j & 1) * RB_R_DW + f_off;
        const float* smr = s_max + (j & 1) * 12 + 8;
        // scale of the staged neighbourhood: a pure function of the data (its max, published by the consumers with the values)
        const float sc = rb_fresh_scale(fmaxf(fmaxf(smr[0], smr[1]), fmaxf(smr[2], smr[3])));
        f32x4 acc[2][2];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) acc[py][px] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            RbFrag F[2][2];
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox) {
                    const float* p = rdl + (oy * 9 + ox) * RB_RPX + 32 * c;      // (pixels outside the image were staged as zeros)
                    F[oy][ox] = rb_split8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), sc);
                }
#define RB_MMA(PY, PX, KY, KX, OY, OX)                                                                                          \
            acc[PY][PX] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c][(KY) * 3 + (KX)], F[OY][OX].l, acc[PY][PX], 0, 0, 0);   \
            acc[PY][PX] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c][(KY) * 3 + (KX)], F[OY][OX].h, acc[PY][PX], 0, 0, 0);   \
            acc[PY][PX] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c][(KY) * 3 + (KX)], F[OY][OX].h, acc[PY][PX], 0, 0, 0);
            // fine (y, x) = (2 Y + ky - 1, 2 X + kx - 1): even y <- ky 1 at Y = y / 2; odd y <- ky 2 at (y - 1) / 2 and ky 0 at (y + 1) / 2
            RB_MMA(0, 0, 1, 1, 0, 0)
            RB_MMA(0, 1, 1, 2, 0, 0) RB_MMA(0, 1, 1, 0, 0, 1)
            RB_MMA(1, 0, 2, 1, 0, 0) RB_MMA(1, 0, 0, 1, 1, 0)
            RB_MMA(1, 1, 2, 2, 0, 0) RB_MMA(1, 1, 2, 0, 0, 1) RB_MMA(1, 1, 0, 2, 1, 0) RB_MMA(1, 1, 0, 0, 1, 1)
#undef RB_MMA
        }
        const float inv = inv_w / sc;
        float md = 0.f;
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const float4 a4 = ax[py][px];
                f32x4 v = acc[py][px] * inv;
                v.x *= elu1_grad_from_out(a4.x); v.y *= elu1_grad_from_out(a4.y);
                v.z *= elu1_grad_from_out(a4.z); v.w *= elu1_grad_from_out(a4.w);
                dv[py][px] = v;
                bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
                md = fmaxf(md, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            }
        if (j + 1 < ntl) ax_load(j + 1);                         // in flight until the end of the next tile's MFMAs
        md = wave_max_f32(md);
        if (lane == 0) s_max[(j & 1) * 12 + 4 + g] = md;
    };

    // pack the raw values of tile j (held in dv since the previous iteration) into plane buffer j & 1
    auto pack = [&](int j) {
        const float* sm = s_max + (j & 1) * 12;
        const float md = fmaxf(fmaxf(sm[4], sm[5]), fmaxf(sm[6], sm[7]));
        sd = tile_scale(md, sd);
        const int buf = j & 1;
        if (ptid == 0) s_sc[buf * 2 + 1] = sd;
        unsigned* pd = s_d + buf * RB_D_DW + d_off;
        // gradient planes: dword (row, p) of channel plane ch = pixels x = 2 p, 2 p + 1 of tile row `row`: the two x-parity classes of
        // one lane are exactly such a pair (x = 2 pj + px), rows py + 2 pi
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned lo;
                const unsigned hi = pack_hi_lo(dv[py][0][e] * sd, dv[py][1][e] * sd, lo);
                pd[e * RB_DPL + py * 8] = hi;
                pd[RB_CO * RB_DPL + e * RB_DPL + py * 8] = lo;
            }
    };

    if (ntl > 0) ax_load(0);
    TP_DECL;
    for (int it = -2; it < ntl; ++it) {
        TP_STAMP(0);
        if (it + 1 >= 0 && it + 1 < ntl) pack(it + 1);
        TP_STAMP(1);                                             // [1] gradient pack -> planes
        if (it + 2 < ntl) fetch_compute(it + 2);
        TP_STAMP(2);                                             // [2] fragments from LDS, MFMAs, ELU', max
        __syncthreads();
        TP_STAMP(3);                                             // [3] barrier
    }
#ifdef IODINE_TILE_PROF
    if (tid == 256 && blockIdx.x < TP_MAXBLK) for (int i_ = 0; i_ < 8; ++i_) g_rb_prof[(TP_MAXBLK + blockIdx.x) * 8 + i_] = tp_acc[i_];
#endif

    // bias gradient partial of this block: sum over the 16 pixel lanes of each channel quad, one row per block
    {
        float b4[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = b4[e];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            b4[e] = v;
        }
        if (n == 0)
            *reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * RB_CO + 16 * g + 4 * cq) = make_float4(b4[0], b4[1], b4[2], b4[3]);
    }
}

}  // namespace

// Fused backward of refinement layers 1 (data gradient) and 0 (weight + bias gradient), split first layer, 64 channels.
//   rd1  [NT][S/4][S/4][64]  gradient wrt the pre-activation of layer 1          wpk / wmeta: launch_pack_conv_weights_ws(W1, 64, tflip 1)
//   act0 [NT][S/2][S/2][64]  saved output of layer 0 (ELU' operand)              enck [NT][S][S][12], encs [NT / kdiv][S][S][8]
// part: nparts x [9][20][64] partial tiles of dW0 in the internal channel order (launch_wgrad_reduce with ci_pad 20),
// part_b: nbias_parts x [64] partial sums of db0.
bool refine_bwd01_ok(int S, int c) { return c == 64 && S >= 64 && (S & (S - 1)) == 0; }      // power-of-two image sizes (tile coordinates are shifts)

hipError_t launch_refine_bwd01(hipStream_t st, const float* rd1, const void* wpk, const float* wmeta, const float* act0, const float* enck,
                               const float* encs, float* part, float* part_b, int NT, int S, int c, int kdiv, int* nparts, int* cipad,
                               int* nbias_parts)
{
    IOD_XSKIP(1024);
    if (!refine_bwd01_ok(S, c) || kdiv < 1) return hipErrorInvalidValue;
    static std::atomic<unsigned> attr_devs{0};
    if (hipError_t e = iod_set_max_lds((const void*)refine_bwd01_kernel, (int)RB_LDS, attr_devs); e != hipSuccess) return e;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    const int Sm = S / 2, tiles_x = Sm / 16, tiles_y = Sm / RB_TH, ntiles = NT * tiles_x * tiles_y;
    int lgSm = 0;
    while ((1 << lgSm) < Sm) ++lgSm;
    // n / kdiv as a multiply-high: validated for every slot-image index of this launch
    const unsigned magic = kdiv == 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + kdiv - 1) / kdiv);     // (2^32 does not fit: 0 = "one slot")
    for (int i = 0; i < NT && kdiv > 1; ++i)
        if ((unsigned)(((unsigned long long)i * magic) >> 32) != (unsigned)(i / kdiv)) return hipErrorInvalidValue;
    const int per_xcd = (ntiles + 7) / 8;
    const int bpx = std::min(per_xcd, std::max(1, n_cu / 8));
    const int blocks = 8 * bpx;
    hipLaunchKernelGGL(refine_bwd01_kernel, dim3(blocks), dim3(512), RB_LDS, st, rd1, reinterpret_cast<const uint4*>(wpk), wmeta, act0,
                       enck, encs, part, part_b, Sm, lgSm, ntiles, magic);
#ifdef IODINE_TILE_PROF
    {
        const int nb = std::min(blocks, TP_MAXBLK);
        std::vector<unsigned> hp((size_t)2 * TP_MAXBLK * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_rb_prof), hp.size() * sizeof(unsigned));
        static const char* cn[8] = {"loop", "enc-pack", "rd-store+load-issue", "frag+mfma", "enc-wait+max", "barrier", "-", "-"};
        static const char* pn[8] = {"loop", "d-pack", "lds-frag+mfma+elu+max", "barrier", "-", "-", "-", "-"};
        for (int role = 0; role < 2; ++role) {
            double sum[8] = {0}, tot = 0;
            for (int b2 = 0; b2 < nb; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[((size_t)role * TP_MAXBLK + b2) * 8 + i];
            for (int i = 0; i < 8; ++i) tot += sum[i] / nb;
            fprintf(stderr, "[refbwd01 prof] %s ticks per block (%d tiles), total %.0f:", role ? "producer" : "consumer", (ntiles + blocks - 1) / blocks, tot);
            for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.0f |", role ? pn[i] : cn[i], sum[i] / nb);
            fprintf(stderr, "\n");
        }
    }
#endif
    *nparts = blocks * 2;
    *cipad = RB_CIR;
    *nbias_parts = blocks;
    return hipGetLastError();
}
