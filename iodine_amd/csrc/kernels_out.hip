// Decoder output conv C -> 4 (rgb pre-sigmoid x3, mask logit; lib/modeling/iodine.py:422,435), streaming form.
//
// Same GEMM formulation as dec_out_gemm_f16x3_kernel (kernels_conv.hip):
//     P[q][tap*4 + co] = sum_ci a[q][ci] * W[co][ci][tap]        one [halo pixels x C] . [C x 36] GEMM per 16x16 tile
//     out[p][co]       = bias[co] + sum_tap P[p + tap - 1][tap*4 + co]
// but the activation operand never goes through LDS: a lane of the 32x32x16 MFMA needs 8 consecutive channels of ONE
// pixel per 16-channel chunk, i.e. 32 contiguous bytes of the NHWC tensor, so every wave fetches its 32-pixel row-blocks
// straight into fragment registers (raw buffer loads, hardware zero outside the image), splits them into fp16 hi/lo in
// registers and multiplies.  No staging buffer, no per-chunk barriers, one row-block of loads in flight behind the one
// being multiplied; the block meets once, when the 324 x 36 P tile is complete.  The old kernel exposed a global-load
// latency per 16-channel chunk (its MFMA phase is 18 instructions per wave): 0.36 ms per launch at cfg3 against an HBM
// floor of ~0.17 ms.  Power-of-two scaling is per row-block (wave-local max over all C channels), so there is no
// accumulator rescale either.  Arithmetic per accumulator: the same three f16 MFMAs per chunk in the same order.
#include "common.h"
#include "pack_bodies.h"
#include <cstdlib>
#include <type_traits>

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_ __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
}  // namespace

// TM: the power-of-two scale of a tile comes from the producer's per-cell max side buffer (tmax: 4 floats per 8 x 16 cell, written by the
// weight-stationary conv / the broadcast layer; max over the 4 x 3 cells the 18 x 18 halo touches) instead of a max over every row-block's
// own values: the kernel is VALU-issue-bound (compute alone 214 us, loads alone 220 us, together 266 us at cfg3: ~300 instructions per
// row-block and wave), and the per-row-block max + wave reduction + scale + 32 multiplies by 1 / scale were 90 of them.  One scale per
// tile also moves the rescale behind the 9-tap sum (4 multiplies per pixel).
template <int C, bool TM>
__global__ __launch_bounds__(256)
void dec_out_stream_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                                 const float* __restrict__ bias, float4* __restrict__ out, int S, int tiles, int ntiles,
                                 const float* __restrict__ tmax)
{
    constexpr int NCHUNK = C / 16;
    constexpr int HALO = 18, NPX = HALO * HALO;                     // 324 halo pixels = 11 row-blocks of 32 (last partial)
    constexpr int W_U4 = NCHUNK * 2 * 2 * 64;                       // [chunk][term hi/lo][kh][64 columns] uint4
    constexpr int PSTR = 36;                                        // floats per pixel in the P tile
    constexpr int NLD = NCHUNK * 2;                                 // float4 loads per lane and row-block

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    uint4* s_w = reinterpret_cast<uint4*>(smem_b);
    float* s_P = reinterpret_cast<float*>(smem_b + W_U4 * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    for (int idx = tid; idx < W_U4; idx += 256) s_w[idx] = wpk[idx];
    const float inv_ws = wmeta[1];
    __syncthreads();                                                // (before the asm loads: hipcc's vmcnt(0) for the copy
                                                                    //  above would otherwise wait for them as well)
    // persistent blocks: tile t, t + gridDim.x, ...; the descriptor and the lane's row-block offsets are per tile
    int tx = 0, ty = 0, n = 0;
    i32x4_ rsrc;
    auto set_tile = [&](int t) {
        tx = t % tiles; t /= tiles;
        ty = t % tiles; n = t / tiles;
        const unsigned long long p = (unsigned long long)(in + (size_t)n * S * S * C);
        rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        rsrc.z = __builtin_amdgcn_readfirstlane(S * S * C * 4);
        rsrc.w = 0x00020000;
    };
    auto rb_offset = [&](int rb) -> unsigned {
        const int px = rb * 32 + li;
        const int gy = ty * 16 - 1 + px / HALO, gx = tx * 16 - 1 + px % HALO;
        const bool ok = px < NPX && gy >= 0 && gy < S && gx >= 0 && gx < S;
        return ok ? (unsigned)(((gy * S + gx) * C + kh * 8) * 4) : 0x80000000u;
    };
    auto issue = [&](unsigned off, f32x4 (&v)[NLD]) {
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            const int soff = c * 64;
            asm volatile("s_nop 4" :: "s"(rsrc), "s"(soff) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v[2 * c]) : "v"(off), "s"(rsrc), "s"(soff) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(v[2 * c + 1]) : "v"(off), "s"(rsrc), "s"(soff) : "memory");
        }
    };
    auto process = [&](int rb, auto nleft, f32x4 (&v)[NLD], float tile_sc) {
        constexpr int nl = decltype(nleft)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(nl) : "memory");
#pragma unroll
        for (int k = 0; k < NLD; ++k) asm volatile("" : "+v"(v[k]));
        __builtin_amdgcn_sched_barrier(0);
        float scale = tile_sc;
        if constexpr (!TM) {
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < NLD; ++k)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w))));
            m = wave_max_f32(m);
            scale = tile_scale(m, 1.f);
        }
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            const f32x4 a = v[2 * c] * scale, b = v[2 * c + 1] * scale;
            unsigned l0, l1, l2, l3;
            const unsigned h0 = pack_hi_lo(a.x, a.y, l0), h1 = pack_hi_lo(a.z, a.w, l1);
            const unsigned h2 = pack_hi_lo(b.x, b.y, l2), h3 = pack_hi_lo(b.z, b.w, l3);
            const u32x4_ uh = {h0, h1, h2, h3}, ul = {l0, l1, l2, l3};
            f16x8 ah, al;
            __builtin_memcpy(&ah, &uh, 16); __builtin_memcpy(&al, &ul, 16);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // (loop-invariant: hipcc keeps these 16 fragments in registers across the persistent tile loop)
                const uint4* q = s_w + ((c * 2 + 0) * 2 + kh) * 64 + nt * 32 + li;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(q);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(q + 2 * 64);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[nt], 0, 0, 0);
            }
        }
        const float inv = TM ? 1.f : inv_ws / scale;         // (TM: rescaled once per pixel behind the tap sum)
        // P rows of this row-block: accumulator register r of lane (li, kh) is pixel (r & 3) + 8 (r >> 2) + 4 kh, column li
        // (+32 for the second column tile, of which only columns 32..35 exist).  Row-block 10 holds pixels 320..323 only.
        float* pb = s_P + (rb * 32 + 4 * kh) * PSTR + li;
        if (rb < 10) {                                              // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[((r & 3) + 8 * (r >> 2)) * PSTR] = acc[0][r] * inv;
            if (li < 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pb[((r & 3) + 8 * (r >> 2)) * PSTR + 32] = acc[1][r] * inv;
            }
        } else if (rb == 10 && kh == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[r * PSTR] = acc[0][r] * inv;
            if (li < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[r * PSTR + 32] = acc[1][r] * inv;
            }
        }
    };

    // wave w owns row-blocks w, w + 4, w + 8 (row-block 11 does not exist: its loads are all out of range, nothing is stored).
    // The first two row-blocks of the NEXT tile are requested before this tile's P tile is summed and stored, so the loads
    // never stop; the output store of the previous tile only makes the counted waits below stricter.
    using std::integral_constant;
    const float4 b4 = make_float4(bias[0], bias[1], bias[2], bias[3]);
    f32x4 va[NLD], vb[NLD];
    // (TM) cell maxima of the tile whose loads are in flight: lane i < 48 holds float i & 3 of cell (2 ty - 1 + i / 12, tx - 1 + (i / 4) % 3),
    // clamped to the image; issued BEFORE the tile's row-blocks, so the counted waits below imply it
    float tmv = 0.f;
    auto issue_tmax = [&]() {
        if constexpr (TM) {
            const int cell = min(lane, 47) >> 2;
            const int cy = min(max(2 * ty - 1 + cell / 3, 0), 2 * tiles - 1), cx = min(max(tx - 1 + cell % 3, 0), tiles - 1);
            const float* p = tmax + (((size_t)n * 2 * tiles + cy) * tiles + cx) * 4 + (lane & 3);
            asm volatile("global_load_dword %0, %1, off" : "=v"(tmv) : "v"(p) : "memory");
        }
    };
    int t = blockIdx.x;
    set_tile(t);
    issue_tmax();
    issue(rb_offset(wv), va);
    issue(rb_offset(wv + 4), vb);
    for (; t < ntiles; t += gridDim.x) {
        const int ctx = tx, cty = ty, cn = n;
        float tsc = 1.f;
        if constexpr (TM) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NLD) : "memory");          // the tile's cell maxima (older than both row-blocks)
            asm volatile("" : "+v"(tmv));
            tsc = tile_scale(wave_max_f32(tmv), 1.f);
        }
        process(wv, integral_constant<int, NLD>{}, va, tsc);
        issue(rb_offset(wv + 8), va);
        process(wv + 4, integral_constant<int, NLD>{}, vb, tsc);
        process(wv + 8, integral_constant<int, 0>{}, va, tsc);
        if (t + (int)gridDim.x < ntiles) {                          // block-uniform
            set_tile(t + gridDim.x);
            issue_tmax();
            issue(rb_offset(wv), va);
            issue(rb_offset(wv + 4), vb);
        }
        __syncthreads();
        const int y = tid >> 4, x = tid & 15;
        float4 o = TM ? make_float4(0.f, 0.f, 0.f, 0.f) : b4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float4 q = *reinterpret_cast<const float4*>(s_P + ((y + tap / 3) * HALO + x + tap % 3) * PSTR + tap * 4);
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        if constexpr (TM) {
            const float inv = inv_ws / tsc;
            o = make_float4(b4.x + o.x * inv, b4.y + o.y * inv, b4.z + o.z * inv, b4.w + o.w * inv);
        }
        out[((size_t)cn * S + cty * 16 + y) * S + ctx * 16 + x] = o;
        __syncthreads();                                            // P tile free for the next tile
    }
}

hipError_t launch_dec_out_stream_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                       const float* bias, float* out, int N, int S, int C, const float* tmax)
{
    IOD_XSKIP(128);
    if (S % 16 != 0) return hipErrorInvalidValue;
    const int tiles = S / 16, ntiles = N * tiles * tiles;
    const int blocks = ntiles < 512 ? ntiles : 512;                 // two resident blocks per CU (63 KB LDS each), persistent
#define DO_LAUNCH(CC, TMF) hipLaunchKernelGGL((dec_out_stream_f16x3_kernel<CC, TMF>), dim3(blocks), dim3(256),                                \
                                              (size_t)(CC / 16) * 2 * 2 * 64 * 16 + 324 * 36 * 4, st, in, reinterpret_cast<const uint4*>(wpk), \
                                              wmeta, bias, reinterpret_cast<float4*>(out), S, tiles, ntiles, tmax)
    if (C == 64) { if (tmax) DO_LAUNCH(64, true); else DO_LAUNCH(64, false); }
    else if (C == 32) { if (tmax) DO_LAUNCH(32, true); else DO_LAUNCH(32, false); }
    else return hipErrorInvalidValue;
#undef DO_LAUNCH
    return hipGetLastError();
}

// =========================================================================================
// Round 5: the same GEMM + 9-tap sum WITHOUT the halo recompute.  The tiled kernel above evaluates P for the 18 x 18 halo of every
// 16 x 16 tile (11 row-blocks of 32 pixels for 8 row-blocks of output: 1.375x the loads, splits and MFMAs) and is VALU-issue-bound on
// exactly that work.  Here a persistent block walks a strip of image rows top to bottom: a STEP is 128 consecutive pixels of the strip
// (one 32-pixel row-block per wave; 128 / S rows), its P values go into a ring of four step slots in LDS (4 x 18 KB), and as soon as
// step j is complete (ONE barrier per step) the outputs of step j - 1 - which need the rows above and below, i.e. steps j - 2 .. j - are
// summed and stored.  Every P row is computed once; a strip of R rows costs R + 2 (128 / S) rows of work (3.6 % at cfg3: 56-row strips).
// Loads are two steps ahead in three register sets (16 KB in flight per wave, 128 KB per CU); the 64 weight fragment registers are
// filled straight from global memory once per block, so LDS holds only the ring (two blocks per CU).  The power-of-two scale of a
// row-block comes from the producer's cell maxima of the two 8 x 16 cells it lies in (tmax, fetched with the row-block) and P is
// stored in true units.  S in {32, 64, 128}; other sizes use the tiled kernel.
// F32 (conv_precision 0): the same pipeline on exact fp32 MFMA - v_mfma_f32_32x32x2_f32, C / 2 k-steps of (2 channels: one per half
// wave) x 2 column tiles, weights as 64 fp32 registers per lane, no scales / side buffer (replaces the round-1 VALU kernel dec_out_kernel:
// 1.25 ms per cfg3 launch).
template <int C, bool F32 = false>
__global__ __launch_bounds__(256, 2)
void dec_out_rows_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                               const float* __restrict__ bias, float* __restrict__ out, int S, int rows_total, int rpb, int nblk,
                               const float* __restrict__ tmax)
{
    constexpr int NCHUNK = C / 16;
    constexpr int PSTR = 36;                                        // floats per pixel of P: (tap, co)
    constexpr int NLD = NCHUNK * 2;                                 // float4 loads per lane and row-block
    constexpr int SLOT = 128 * PSTR;                                // floats per ring slot (one step)
    extern __shared__ __attribute__((aligned(16))) float s_ring[];  // [4][128][PSTR]

    const int tid = threadIdx.x;
    const int lane = tid & 63, kh = lane >> 5, li = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int SPR = 128 / S;                                        // rows per step
    // XCD-aware: block b runs on XCD b % 8; every XCD gets one contiguous eighth of the row ranges (neighbours share a halo row in L2)
    const int per_xcd = (nblk + 7) >> 3;
    const int vb = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (vb >= nblk) return;
    const int r0 = vb * rpb, r1 = min(rows_total, r0 + rpb);

    // weight fragments -> registers (loop-invariant): lane (li, kh), column tile nt, chunk c, hi / lo
    constexpr int NOPS = NLD + (F32 ? 0 : 1);                      // memory operations per issue(): the row-block (+ its cell maxima)
    f16x8 wh[F32 ? 1 : NCHUNK][2], wl[F32 ? 1 : NCHUNK][2];
    float wf[F32 ? C / 2 : 1][2];                                  // F32: B operand of k-step s, column tile nt (pack_dec_out_rows32_kernel)
    if constexpr (F32) {
        const float* wp = reinterpret_cast<const float*>(wpk);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < C / 2; ++ks) wf[ks][nt] = wp[(nt * (C / 2) + ks) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < C / 2; ++ks) asm volatile("" : "+v"(wf[ks][nt]));
    } else {
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const uint4 a = wpk[((c * 2 + 0) * 2 + kh) * 64 + nt * 32 + li], b = wpk[((c * 2 + 1) * 2 + kh) * 64 + nt * 32 + li];
            __builtin_memcpy(&wh[c][nt], &a, 16); __builtin_memcpy(&wl[c][nt], &b, 16);
        }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) asm volatile("" : "+v"(wh[c][nt]), "+v"(wl[c][nt]));
    }
    float inv_ws = 1.f;
    if constexpr (!F32) inv_ws = wmeta[1];
    float2 b2 = make_float2(bias[2 * (tid & 1)], bias[2 * (tid & 1) + 1]);            // this thread's channel pair of the output
    // (pinned here: hipcc's own wait for these two loads would otherwise be a vmcnt(0) at their first use INSIDE the pipelined loop)
    asm volatile("" : "+v"(inv_ws), "+v"(b2.x), "+v"(b2.y));

    f32x4 v0[NLD], v1[NLD], v2[NLD];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    i32x4_ rsrc;
    const int tcx = S >> 4, tcy = S >> 3;                           // cells per row / column of the side buffer

    for (int ra = r0; ra < r1;) {
        // ---- sub-strip [ya, yb) of image n (a block's row range is cut at image boundaries) ----
        const int n = ra / S, ya = ra - n * S;
        const int yb = min(S, ya + (r1 - ra));
        ra += yb - ya;
        const int G = (yb - ya) / SPR, J = G + 2;                   // output groups, P steps (one group of rows above, one below)
        {
            const unsigned long long p = (unsigned long long)(in + (size_t)n * S * S * C);
            rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
            rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
            rsrc.z = __builtin_amdgcn_readfirstlane(S * S * C * 4);
            rsrc.w = 0x00020000;
        }
        const int ystart = ya - SPR;                                // image row of pixel 0 of step 0 (may be negative: zeros)
        // loads of (step j, this wave's row-block) + the two cell maxima its scale comes from: NLD + 1 memory operations
        auto issue = [&](int j, f32x4 (&v)[NLD], float& tm) {
            const int q0 = j * 128 + 32 * wv;                       // first pixel of the row-block (wave-uniform)
            const int row = ystart + q0 / S, x = q0 % S + li;
            const bool ok = row >= 0 && row < S;
            const unsigned off = ok ? (unsigned)(((row * S + x) * C + kh * 8) * 4) : 0x80000000u;
            const int crow = min(max(row, 0), S - 1) >> 3, ccol = (q0 % S) >> 4;
            if constexpr (!F32) {
                const float* tp = tmax + (((size_t)n * tcy + crow) * tcx + ccol) * 4 + (lane & 7);
                asm volatile("global_load_dword %0, %1, off" : "=v"(tm) : "v"(tp) : "memory");
            }
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {
                const int soff = c * 64;
                asm volatile("s_nop 4" :: "s"(rsrc), "s"(soff) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v[2 * c]) : "v"(off), "s"(rsrc), "s"(soff) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(v[2 * c + 1]) : "v"(off), "s"(rsrc), "s"(soff) : "memory");
            }
        };
        // P of (step j, this wave's row-block) -> ring slot j & 3
        auto process = [&](int j, f32x4 (&v)[NLD], float& tm) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) asm volatile("" : "+v"(v[k]));
            asm volatile("" : "+v"(tm));
            __builtin_amdgcn_sched_barrier(0);
            float scale = 1.f;
            if constexpr (!F32) scale = tile_scale(wave_max_f32(lane < 8 ? tm : 0.f), 1.f);
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            if constexpr (F32) {
                // k-step ks: this lane's channel 16 (ks / 8) + 8 kh + ks % 8 = element ks of its NLD float4 (the loads' own order)
#pragma unroll
                for (int ks = 0; ks < C / 2; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[ks / 4][ks % 4], wf[ks][nt], acc[nt], 0, 0, 0);
            } else
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {
                const f32x4 a = v[2 * c] * scale, b = v[2 * c + 1] * scale;
                unsigned l0, l1, l2, l3;
                const unsigned h0 = pack_hi_lo(a.x, a.y, l0), h1 = pack_hi_lo(a.z, a.w, l1);
                const unsigned h2 = pack_hi_lo(b.x, b.y, l2), h3 = pack_hi_lo(b.z, b.w, l3);
                const u32x4_ uh = {h0, h1, h2, h3}, ul = {l0, l1, l2, l3};
                f16x8 ah, al;
                __builtin_memcpy(&ah, &uh, 16); __builtin_memcpy(&al, &ul, 16);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[c][nt], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[c][nt], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[c][nt], acc[nt], 0, 0, 0);
                }
            }
            const float inv = inv_ws / scale;
            // accumulator register r of lane (li, kh) is pixel (r & 3) + 8 (r >> 2) + 4 kh of the row-block, column li (+ 32: columns 32 .. 35)
            float* pb = s_ring + (j & 3) * SLOT + (32 * wv + 4 * kh) * PSTR + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[((r & 3) + 8 * (r >> 2)) * PSTR] = acc[0][r] * inv;
            if (li < 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pb[((r & 3) + 8 * (r >> 2)) * PSTR + 32] = acc[1][r] * inv;
            }
        };
        // outputs of group g (its pixels are those of P step g + 1): thread = (pixel tid / 2, channel pair tid & 1)
        auto emit = [&](int g) {
            const int i = tid >> 1, cp = tid & 1;
            const int y = ya + g * SPR + i / S, x = i % S;
            const int q = (g + 1) * 128 + i;
            float2 o = b2;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                if ((unsigned)(x + dx) < (unsigned)S) {
                    const int qq = q + dy * S + dx;
                    const float2 pv = *reinterpret_cast<const float2*>(s_ring + ((qq >> 7) & 3) * SLOT + (qq & 127) * PSTR + tap * 4 + cp * 2);
                    o.x += pv.x; o.y += pv.y;
                }
            }
            float2* op = reinterpret_cast<float2*>(out + (((size_t)n * S + y) * S + x) * 4 + cp * 2);
            asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" :: "v"(op), "v"(o) : "memory");
        };
        // ---- software pipeline: loads two steps ahead (three register sets), one barrier per step ----
        // vmcnt is in order: behind the loads of step j sit [store of emit(j - 4)] [loads j + 1] [store of emit(j - 3)] [loads j + 2];
        // the stores do not exist in the first steps of a strip, so the counts below allow for the two younger load groups only
        // (at worst an old 8-byte store is waited for as well)
        auto stepf = [&](int j, f32x4 (&cur)[NLD], float& tcur, f32x4 (&nx2)[NLD], float& tnx2) {
            if (j + 2 < J) {
                issue(j + 2, nx2, tnx2);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NOPS) : "memory");
            } else if (j + 1 < J) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NOPS) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            process(j, cur, tcur);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS-only barrier: the prefetch stays in flight
            if (j >= 2) emit(j - 2);
        };
        issue(0, v0, t0);
        issue(1, v1, t1);
        int j = 0;
        for (; j + 2 < J; j += 3) {
            stepf(j, v0, t0, v2, t2);
            stepf(j + 1, v1, t1, v0, t0);
            stepf(j + 2, v2, t2, v1, t1);
        }
        if (j < J) { stepf(j, v0, t0, v2, t2); ++j; }
        if (j < J) { stepf(j, v1, t1, v0, t0); ++j; }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // the ring is free for the next sub-strip
    }
}

bool dec_out_rows_ok(int S, int C, const float* tmax) { return tmax && (S == 32 || S == 64 || S == 128) && (C == 64 || C == 32); }

// fp32 weights of the F32 form: dst[(nt * (C / 2) + ks) * 64 + lane] = W[column nt * 32 + lane % 32 (= tap * 4 + co, zero from 36)][channel of
// (k-step ks, half wave lane / 32) = 16 (ks / 8) + 8 (lane / 32) + ks % 8]
__global__ void pack_dec_out_rows32_kernel(const float* __restrict__ w /*[4][C][3][3]*/, int C, float* __restrict__ dst)
{
    const int total = 2 * (C / 2) * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, ks = (idx >> 6) % (C / 2), nt = (idx >> 6) / (C / 2);
        const int j = nt * 32 + (lane & 31), ci = 16 * (ks / 8) + 8 * (lane >> 5) + ks % 8;
        dst[idx] = j < 36 ? w[((size_t)(j & 3) * C + ci) * 9 + (j >> 2)] : 0.f;
    }
}

hipError_t launch_pack_dec_out_rows32(hipStream_t st, const float* w, int C, float* dst)
{
    hipLaunchKernelGGL(pack_dec_out_rows32_kernel, dim3(C / 2), dim3(128), 0, st, w, C, dst);
    return hipGetLastError();
}

hipError_t launch_dec_out_rows_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias, float* out,
                                     int N, int S, int C, const float* tmax, int f32)
{
    IOD_XSKIP(128);
    if (!dec_out_rows_ok(S, C, f32 ? in : tmax)) return hipErrorInvalidValue;      // (the fp32 form needs no side buffer)
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    const int spr = 128 / S, rows_total = N * S;
    const int want = 2 * n_cu;                                      // two resident blocks per CU (72 KB of LDS each)
    int rpb = (rows_total + want - 1) / want;
    rpb = (rpb + spr - 1) / spr * spr;
    const int nblk = (rows_total + rpb - 1) / rpb;
    const int grid = ((nblk + 7) / 8) * 8;
    constexpr size_t lds = (size_t)4 * 128 * 36 * 4;
#define DO_ROWS(CC, FF)                                                                                                                     \
    {                                                                                                                                       \
        static std::atomic<unsigned> attr_devs{0};                                                                                          \
        if (hipError_t e = iod_set_max_lds((const void*)dec_out_rows_f16x3_kernel<CC, FF>, (int)lds, attr_devs); e != hipSuccess) return e; \
        hipLaunchKernelGGL((dec_out_rows_f16x3_kernel<CC, FF>), dim3(grid), dim3(256), lds, st, in, reinterpret_cast<const uint4*>(wpk),    \
                           wmeta, bias, out, S, rows_total, rpb, nblk, tmax);                                                               \
    }
    if (C == 64) { if (f32) DO_ROWS(64, true) else DO_ROWS(64, false) }
    else { if (f32) DO_ROWS(32, true) else DO_ROWS(32, false) }
#undef DO_ROWS
    return hipGetLastError();
}

// =========================================================================================
// Data gradient of the output conv (4 -> C channels, times ELU' of the saved activation): the first kernel of every
// decoder backward pass.  out[p][c] = ELU'(aux[p][c]) * sum_{tap, o} g[p + tap - 1][o] * W[o][c][8 - tap]: a
// [pixels x 36] . [36 x C] GEMM (K = tap*4 + o, padded to 48) whose A operand comes from the 4-channel gradient of the
// decoder output (5 KB halo tile per block) - the launch is a stream of aux in and out out (1.9 GB at cfg3).  The generic
// exact-fp32 tile kernel it replaces spent 72 fp32 MFMAs (4.6 k matrix-pipe cycles) per wave and tile and moved the big
// tensors with dword loads / stores; this one issues 36 split-fp16 MFMAs, fetches the ELU' operand as whole pixels at
// block start (it lands under the staging and the MFMAs) and writes whole pixels after a transposition through LDS.
//   wpk: [chunk 3][term hi/lo][kh 2][C] uint4 = 8 fp16 (k = chunk*16 + kh*8 + e; k = tap*4 + o, zero for k >= 36) * wscale
// =========================================================================================
__global__ void pack_dec_out_dgrad_kernel(const float* __restrict__ w /*[4][C][3][3]*/, int C, const float* __restrict__ meta,
                                          _Float16* __restrict__ dst)
{
    const float scale = meta[0];
    const int total = (int)pack_out_dgrad_total(C);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
        dst[idx] = pack_out_dgrad_element(w, C, scale, (size_t)idx);         // (pack_bodies.h: shared with the batched form)
}

hipError_t launch_pack_dec_out_dgrad(hipStream_t st, const float* w, int C, const float* meta, void* dst)
{
    hipLaunchKernelGGL(pack_dec_out_dgrad_kernel, dim3(12), dim3(256), 0, st, w, C, meta, (_Float16*)dst);
    return hipGetLastError();
}

template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void dec_out_dgrad_f16x3_kernel(const float4* __restrict__ g, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                                const float* __restrict__ aux, float* __restrict__ out, float* __restrict__ tmax, int S, int tiles)
{
    constexpr int NT = C / 32;
    constexpr int HALO = 18, NPX = HALO * HALO;
    constexpr int W_U4 = 3 * 2 * 2 * C;
    constexpr int EPS = (C + 4) * 4;                                // bytes per transposed pixel
    constexpr int SEGS = C / 4, PPI = 64 / SEGS, NEP = 32 / PPI;    // per 32-pixel half tile of a wave

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    float4* s_g = reinterpret_cast<float4*>(smem_b);                // [324] gradient halo tile, fp32
    uint4* s_w = reinterpret_cast<uint4*>(smem_b + NPX * 16);       // packed weights
    float* s_max = reinterpret_cast<float*>(smem_b + NPX * 16 + W_U4 * 16);
    unsigned char* s_ep = smem_b + NPX * 16 + W_U4 * 16 + 16;       // [4 waves][32 px][EPS]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    int bid = blockIdx.x;
    const int tx = bid % tiles; bid /= tiles;
    const int ty = bid % tiles;
    const int n = bid / tiles;

    // ELU' operand: whole pixels (lane = pixel pl of PPI, 16-byte segment seg), requested first
    const int seg = lane % SEGS, pl = lane / SEGS;
    const float* aux_n = aux + (size_t)n * S * S * C;
    float* out_n = out + (size_t)n * S * S * C;
    f32x4 ax[2][NEP];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < NEP; ++j) {
            const int pq = j * PPI + pl;                            // pixel of the half tile: row pq / 16, column pq % 16
            const int gy = ty * 16 + 4 * wv + 2 * mt + pq / 16, gx = tx * 16 + pq % 16;
            const float4 t = *reinterpret_cast<const float4*>(aux_n + ((size_t)gy * S + gx) * C + seg * 4);
            ax[mt][j] = f32x4{t.x, t.y, t.z, t.w};
        }
    // gradient halo tile + weights -> LDS, block max of |g|
    float m = 0.f;
    for (int idx = tid; idx < NPX; idx += 256) {
        const int gy = ty * 16 - 1 + idx / HALO, gx = tx * 16 - 1 + idx % HALO;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < S && gx >= 0 && gx < S) v = g[((size_t)n * S + gy) * S + gx];
        s_g[idx] = v;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int idx = tid; idx < W_U4; idx += 256) s_w[idx] = wpk[idx];
    m = wave_max_f32(m);
    if (lane == 0) s_max[wv] = m;
    __syncthreads();
    const float scale = tile_scale(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])), 1.f);
    const float inv = wmeta[1] / scale;

    unsigned char* ep = s_ep + wv * 32 * EPS;
    float omax = 0.f;                                               // max |output| of this wave's four tile rows
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        // accumulators of this half tile: rows = channels (weights are the first MFMA operand), columns = 32 pixels
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int py = 4 * wv + 2 * mt + (li >> 4), px = li & 15;  // tile pixel of this lane's fragment column
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int t0 = 4 * c + 2 * kh;                          // this lane's two taps of the chunk
            float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
            if (t0 < 9) q0 = s_g[(py + t0 / 3) * HALO + px + t0 % 3];
            if (t0 + 1 < 9) q1 = s_g[(py + (t0 + 1) / 3) * HALO + px + (t0 + 1) % 3];
            unsigned l0, l1, l2, l3;
            const unsigned h0 = pack_hi_lo(q0.x * scale, q0.y * scale, l0), h1 = pack_hi_lo(q0.z * scale, q0.w * scale, l1);
            const unsigned h2 = pack_hi_lo(q1.x * scale, q1.y * scale, l2), h3 = pack_hi_lo(q1.z * scale, q1.w * scale, l3);
            const u32x4_ uh = {h0, h1, h2, h3}, ul = {l0, l1, l2, l3};
            f16x8 ah, al;
            __builtin_memcpy(&ah, &uh, 16); __builtin_memcpy(&al, &ul, 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const uint4* q = s_w + ((c * 2 + 0) * 2 + kh) * C + nt * 32 + li;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(q);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(q + 2 * C);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[nt], 0, 0, 0);
            }
        }
        // transposition through the wave's LDS region: lane (li, kh) holds channels nt*32 + 8*g4 + 4*kh .. +3 of pixel li
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<f32x4*>(ep + li * EPS + (nt * 32 + 8 * g4 + 4 * kh) * 4) =
                    f32x4{acc[nt][4 * g4] * inv, acc[nt][4 * g4 + 1] * inv, acc[nt][4 * g4 + 2] * inv, acc[nt][4 * g4 + 3] * inv};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < NEP; ++j) {
            const int pq = j * PPI + pl;
            const int gy = ty * 16 + 4 * wv + 2 * mt + pq / 16, gx = tx * 16 + pq % 16;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + pq * EPS + seg * 16);
            const f32x4 a4 = ax[mt][j];
            v.x *= elu1_grad_from_out(a4.x); v.y *= elu1_grad_from_out(a4.y);
            v.z *= elu1_grad_from_out(a4.z); v.w *= elu1_grad_from_out(a4.w);
            omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            *reinterpret_cast<f32x4*>(out_n + ((size_t)gy * S + gx) * C + seg * 4) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (tmax) {
        // side buffer for the weight-stationary conv that consumes `out`: max |x| per 8 x 16 cell, 4 floats per cell; the
        // 16 x 16 tile is two cells (waves 0, 1 / 2, 3), each wave fills two of its cell's four slots
        omax = wave_max_f32(omax);
        if (lane < 2) tmax[(((size_t)n * 2 * tiles + 2 * ty + (wv >> 1)) * tiles + tx) * 4 + 2 * (wv & 1) + lane] = omax;
    }
}

hipError_t launch_dec_out_dgrad_f16x3(hipStream_t st, const float* g, const void* wpk, const float* wmeta, const float* aux,
                                      float* out, int N, int S, int C, float* tmax)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
    const int tiles = S / 16;
    if (C == 64) {
        constexpr size_t lds = (size_t)324 * 16 + 3 * 2 * 2 * 64 * 16 + 16 + 4 * 32 * (64 + 4) * 4;
        hipLaunchKernelGGL((dec_out_dgrad_f16x3_kernel<64>), dim3(N * tiles * tiles), dim3(256), lds, st, (const float4*)g,
                           (const uint4*)wpk, wmeta, aux, out, tmax, S, tiles);
    } else if (C == 32) {
        constexpr size_t lds = (size_t)324 * 16 + 3 * 2 * 2 * 32 * 16 + 16 + 4 * 32 * (32 + 4) * 4;
        hipLaunchKernelGGL((dec_out_dgrad_f16x3_kernel<32>), dim3(N * tiles * tiles), dim3(256), lds, st, (const float4*)g,
                           (const uint4*)wpk, wmeta, aux, out, tmax, S, tiles);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
