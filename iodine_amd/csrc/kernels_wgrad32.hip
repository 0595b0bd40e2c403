// Exact-fp32 weight gradient of the decoder's stride-1 3x3 conv C -> C (option conv_precision 0), persistent and
// software-pipelined:  dW[tap][ci][co] = sum_{n, p} a[n, p + tap - 1, ci] * d[n, p, co]   (what autograd computes for nn.Conv2d,
// lib/modeling/iodine.py:583, in the outer loss.backward() of lib/engine/train.py:63).
//
// GEMM view per tap: M = ci, N = co, K = pixels, v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate; 64 cycles per MFMA
// and SIMD = the fp32 matrix peak of 157.3 TF/s when the pipe never idles).  Both operands are K-major in NHWC - lane (half, li) of a
// k-step reads channel li of pixel 2 s + half: no transposition, one ds_read_b32 per operand.  The round-1 kernel
// (conv3x3_wgrad_tile_kernel, kernels_train.hip) loaded a tile, synchronised, multiplied, synchronised: its global loads were fully
// exposed (0.70 of the peak).  Here
//   * a wave keeps the nine 32 x 32 tap accumulators of its (ci half, co half) in 144 VGPRs for the life of a persistent block;
//   * the NEXT 4 x 16 tile (6 x 18 halo of a, 4 x 16 of d) is fetched into registers (raw buffer loads: halo pixels outside the
//     image come back as 0 from the bounds check of the per-slot-image descriptor) while the 288 MFMAs of the current tile run from
//     LDS; two barriers and eleven ds_write_b128 per thread separate two tiles, and the second block of the CU (2 x 44 KB of LDS,
//     2 x ~210 VGPRs) fills the matrix pipe meanwhile;
//   * LDS planes of 32 channels ([ci half][pixel][32]): the two half-waves of a ds_read_b32 hit disjoint bank halves;
//   * XCD-aware schedule: every XCD walks one contiguous eighth of the tile list (halo re-reads stay in its L2).
// Partial tiles part[(block * KS + ks)][tap][ci][co] are reduced in fixed order by launch_wgrad_reduce (deterministic).
#include "common.h"
#include <type_traits>

namespace {

typedef int i32x4w_ __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
IOD_DEVINL void w32_static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w32_static_for<I + 1, N>(f);
    }
}

}  // namespace

template <int C>
__global__ __launch_bounds__(256, 2)
void conv3x3_wgrad_f32_ws_kernel(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ part,
                                 float* __restrict__ part_b, int S, int ntiles, float alpha, int accum)
{
    constexpr int MT = C / 32;                   // ci halves
    constexpr int NTT = C / 32;                  // co halves
    constexpr int KS = 4 / (MT * NTT);           // pixel split over the waves (C = 32: one tile row per wave)
    constexpr int TH = 4, TW = 16, HW = TW + 2, NHALO = (TH + 2) * HW, NPXT = TH * TW;
    constexpr int Q = C / 4;                     // float4 per pixel
    constexpr int NA = (NHALO * Q + 255) / 256;  // float4 loads per thread and tile: halo of a (7 at C = 64) ...
    constexpr int ND = NPXT * Q / 256;           // ... and d (4)
    constexpr int PLANE_A = NHALO * 128, PLANE_D = NPXT * 128;    // bytes of one 32-channel plane
    constexpr int STEPS = NPXT / KS / 2;         // k-steps (pixel pairs) per wave and tile
    static_assert(C == 64 || C == 32, "channel counts of the shipped decoders");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w32[];
    unsigned char* s_a = smem_w32;                       // [MT][NHALO][32] floats
    unsigned char* s_d = smem_w32 + MT * PLANE_A;        // [NTT][NPXT][32] floats

    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv % MT, ni = (wv / MT) % NTT, ks = wv / (MT * NTT);

    auto make_rsrc = [&](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)base;
        i32x4w_ r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
#define W32_BLOAD4(dst, voff, rsrc, soff) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory")
#define W32_SGPR_SETTLE(rsrc) asm volatile("s_nop 4" :: "s"(rsrc) : "memory")

    // persistent, XCD-aware schedule (block b runs on XCD b % 8: observed, used for locality only)
    const int nblk = gridDim.x;
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = (nblk + 7) >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_begin = xcd * per_xcd, t_end = min(ntiles, t_begin + per_xcd);
    const int tiles_x = S / TW, tiles_y = S / TH;

    // staging roles: float4 q4 = tid % Q of pixel tid / Q + (256 / Q) k
    const int q4 = tid % Q, p0 = tid / Q;
    constexpr int PSTEP = 256 / Q;
    const unsigned lw_a = (unsigned)((q4 >> 3) * PLANE_A + (q4 & 7) * 16), lw_d = (unsigned)((q4 >> 3) * PLANE_D + (q4 & 7) * 16);

    // per-thread constants of the staging loads: byte offset of halo pixel k relative to the halo origin (ty * 4 - 1, tx * 16 - 1) and
    // which image borders would put it outside (4 bits per k); d: pixel p0 + PSTEP k = row (PSTEP / 16) k + p0 / 16, column p0 % 16 -
    // the k term goes into the scalar offset of the load
    unsigned rel_a[NA], bmask = 0;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int px = p0 + PSTEP * k;
        const int hy = px / HW, hx = px - hy * HW;
        rel_a[k] = px < NHALO ? (unsigned)(((hy * S + hx) * C + q4 * 4) * 4) : 0xC0000000u;      // idle lanes of the last load: never in range
        bmask |= (unsigned)((hy == 0 ? 1 : 0) | (hy == TH + 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == HW - 1 ? 8 : 0)) << (4 * k);
    }
    const unsigned rel_d = (unsigned)((((p0 / TW) * S + p0 % TW) * C + q4 * 4) * 4);
    const int d_kstride = (PSTEP / TW) * S * C * 4;

    f32x4 ra[NA], rd[ND];
    auto issue_tile = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const unsigned img_bytes = (unsigned)(S * S * C * 4);
        const i32x4w_ rs_a = make_rsrc(a + (size_t)n * S * S * C, img_bytes);
        const i32x4w_ rs_d = make_rsrc(d + (size_t)n * S * S * C, img_bytes);
        // (scalar) borders this tile touches, replicated for every k; byte offset of the halo origin (may be "negative": wraps)
        const unsigned tb = (ty == 0 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == tiles_x - 1 ? 8u : 0u);
        const unsigned out = bmask & (tb * 0x11111111u);
        const unsigned org_a = (unsigned)((((ty * TH - 1) * S + tx * TW - 1) * C) * 4);
        const unsigned org_d = (unsigned)((((ty * TH) * S + tx * TW) * C) * 4);
        unsigned oa[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) oa[k] = ((out >> (4 * k)) & 0xfu) ? 0x80000000u : rel_a[k] + org_a;     // out of range: the load returns 0
        const unsigned od = rel_d + org_d;
        W32_SGPR_SETTLE(rs_a);
#pragma unroll
        for (int k = 0; k < NA; ++k) W32_BLOAD4(ra[k], oa[k], rs_a, 0);
        W32_SGPR_SETTLE(rs_d);
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int soff = k * d_kstride;
            asm volatile("s_nop 4" :: "s"(soff) : "memory");
            W32_BLOAD4(rd[k], od, rs_d, soff);
        }
    };
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
    auto land_tile = [&]() {                              // registers -> LDS (+ this thread's share of the bias gradient)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NA; ++k) asm volatile("" : "+v"(ra[k]));
#pragma unroll
        for (int k = 0; k < ND; ++k) asm volatile("" : "+v"(rd[k]));
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int px = p0 + PSTEP * k;
            if (k < NA - 1 || px < NHALO) *reinterpret_cast<f32x4*>(s_a + lw_a + (unsigned)px * 128) = ra[k];
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            *reinterpret_cast<f32x4*>(s_d + lw_d + (unsigned)(p0 + PSTEP * k) * 128) = rd[k];
            bsum += rd[k];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int t0 = t_begin + bix;
    if (t0 < t_end) {
        issue_tile(t0);
        land_tile();
        __syncthreads();
        // operand addresses of this lane: pixel (ks * rows + ...) + half, channel li of plane mi / ni
        const float* ap = reinterpret_cast<const float*>(s_a + mi * PLANE_A) + (ks * (TH / KS) * HW + half) * 32 + li;
        const float* dp = reinterpret_cast<const float*>(s_d + ni * PLANE_D) + (ks * (NPXT / KS) + half) * 32 + li;
        for (int t = t0; t < t_end; t += bpx) {
            const bool has_next = t + bpx < t_end;
            if (has_next) issue_tile(t + bpx);
            // ---- 9 x STEPS MFMAs of this tile from LDS: straight-line code, literal LDS offsets ----
            w32_static_for<0, STEPS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int r = (2 * s) / TW, c = (2 * s) % TW;
                const float bval = dp[(2 * s) * 32];
                float av[9];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) av[tap] = ap[((r + tap / 3) * HW + c + tap % 3) * 32];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tap], bval, acc[tap], 0, 0, 0);
            });
            if (has_next) {
                __syncthreads();                          // every wave is done reading this tile
                land_tile();
                __syncthreads();
            }
        }
    }

    // partial dW: rows = ci (accumulator rows), cols = co (lane & 31)
    // (accum / alpha: the block's partial tile accumulates alpha x this launch over the decoder passes of a training step - one
    // fixed-order reduction per layer and step, see conv3x3_wgrad_f16x3_ws_kernel)
    constexpr int NCOP = NTT * 32;
    float* pw = part + ((size_t)(blockIdx.x * KS + ks) * 9) * C * NCOP;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = accum ? pw[((size_t)tap * C + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * NCOP + ni * 32 + li] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            pw[((size_t)tap * C + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * NCOP + ni * 32 + li] = old[r] + alpha * acc[tap][r];
    }
    // partial bias gradient: a thread's channel quad is fixed (256 % Q == 0); fixed-order sum over the threads that share it
    __syncthreads();
    f32x4* s_red = reinterpret_cast<f32x4*>(smem_w32);
    s_red[tid] = bsum;
    __syncthreads();
    if (tid < Q) {
        f32x4 t4 = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = tid; j < 256; j += Q) t4 += s_red[j];
        f32x4* pb = reinterpret_cast<f32x4*>(part_b + (size_t)blockIdx.x * C + tid * 4);
        f32x4 o4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (accum) o4 = *pb;
        *pb = o4 + t4 * alpha;
    }
#undef W32_BLOAD4
#undef W32_SGPR_SETTLE
}

int wgrad_f32_ws_blocks(int N, int S, int n_cu)
{
    const int nt = N * (S / 4) * (S / 16);
    const int per_xcd = (nt + 7) / 8;
    const int bpx = std::min(per_xcd, std::max(1, 2 * n_cu / 8));       // two persistent blocks per CU
    return 8 * bpx;
}

template <int C>
static hipError_t launch_wgrad_f32_ws_inst(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N, int S,
                                           int* nparts, int* ncop, int* nbias_parts, float alpha, int accum)
{
    constexpr int KS = 4 / ((C / 32) * (C / 32));
    constexpr size_t lds = (size_t)(C / 32) * (6 * 18 + 4 * 16) * 128;
    static std::atomic<unsigned> attr_devs{0};
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_wgrad_f32_ws_kernel<C>, (int)lds, attr_devs); e != hipSuccess) return e;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    const int ntiles = N * (S / 4) * (S / 16);
    int blocks = wgrad_f32_ws_blocks(N, S, n_cu);
    if (blocks > 512) blocks = 512;                                      // capacity of the partial-tile buffers (plan(): 512 blocks)
    hipLaunchKernelGGL((conv3x3_wgrad_f32_ws_kernel<C>), dim3(blocks), dim3(256), lds, st, a, d, part, part_b, S, ntiles, alpha, accum);
    *nparts = blocks * KS;
    *ncop = C;
    *nbias_parts = blocks;
    return hipGetLastError();
}

// a, d: NHWC [N][S][S][c]; part: [nparts][9][c][c], part_b: [nbias_parts][c] (reduced by launch_wgrad_reduce)
hipError_t launch_conv3x3_wgrad_f32_ws(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N, int S, int c,
                                       int* nparts, int* ncop, int* nbias_parts, float alpha, int accum)
{
    if (S % 16 != 0 || (size_t)S * S * c * 4 >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    if (c == 64) return launch_wgrad_f32_ws_inst<64>(st, a, d, part, part_b, N, S, nparts, ncop, nbias_parts, alpha, accum);
    if (c == 32) return launch_wgrad_f32_ws_inst<32>(st, a, d, part, part_b, N, S, nparts, ncop, nbias_parts, alpha, accum);
    return hipErrorInvalidValue;
}

// =========================================================================================
// Exact-fp32 weight gradient of the OUTPUT conv C -> 4 (conv_precision 0) in GEMM form:
//     dW[tap][ci][co] = sum_q a[q][ci] * g[q - (tap - centre)][co]          rows ci, columns j = tap * 4 + co (36 of 64), K = pixels
// - the tap shift is moved onto the 4-channel gradient g (a 6 x 18 halo tile of float4, 1.7 KB), the 64-channel operand needs no halo
// and is read exactly once: 4 MFMAs (v_mfma_f32_32x32x2_f32) per pixel pair and block where the direct form with N = 4 padded to 32
// (conv3x3_wgrad_tile_kernel<C, 4>, round 1) issues 18 - 1.29 ms per cfg3 launch for 0.94 GB of input.  Same persistent / prefetched
// structure as conv3x3_wgrad_f32_ws_kernel above; partial tiles part[block * KS + ks][tap][ci][4], part_b[block][4].
// =========================================================================================
template <int C>
__global__ __launch_bounds__(256, 4)
void dec_out_wgrad_f32_kernel(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ part, float* __restrict__ part_b,
                              int S, int ntiles)
{
    constexpr int MT = C / 32, KS = 2 / MT;              // waves = MT (ci halves) x 2 (column tiles) x KS (pixel split)
    constexpr int TH = 4, TW = 16, HW = TW + 2, NHALO = (TH + 2) * HW, NPXT = TH * TW;
    constexpr int Q = C / 4, NA = NPXT * Q / 256, PSTEP = 256 / Q;
    constexpr int PLANE_A = NPXT * 128;
    constexpr int STEPS = NPXT / KS / 2;
    static_assert(C == 64 || C == 32, "channel counts of the shipped decoders");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ow[];
    unsigned char* s_a = smem_ow;                                      // [MT][64 px][32] floats
    float* s_g = reinterpret_cast<float*>(smem_ow + MT * PLANE_A);     // [NHALO][4] floats

    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv % MT, nt = (wv / MT) & 1, ks = wv / (MT * 2);

    auto make_rsrc = [&](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)base;
        i32x4w_ r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
#define OW_BLOAD4(dst, voff, rsrc, soff) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory")

    const int nblk = gridDim.x;
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = (nblk + 7) >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_begin = xcd * per_xcd, t_end = min(ntiles, t_begin + per_xcd);
    const int tiles_x = S / TW, tiles_y = S / TH;

    // staging roles: a - float4 q4 of pixel p0 + PSTEP k (row (PSTEP / 16) k + p0 / 16, column p0 % 16); g - thread < NHALO: one halo pixel
    const int q4 = tid % Q, p0 = tid / Q;
    const unsigned lw_a = (unsigned)((q4 >> 3) * PLANE_A + (q4 & 7) * 16);
    const unsigned rel_a = (unsigned)((((p0 / TW) * S + p0 % TW) * C + q4 * 4) * 4);
    const int a_kstride = (PSTEP / TW) * S * C * 4;
    const int ghy = tid / HW, ghx = tid - ghy * HW;
    const bool g_role = tid < NHALO;
    const bool g_inner = g_role && ghy >= 1 && ghy <= TH && ghx >= 1 && ghx <= TW;
    const unsigned gmask = (unsigned)((ghy == 0 ? 1 : 0) | (ghy == TH + 1 ? 2 : 0) | (ghx == 0 ? 4 : 0) | (ghx == HW - 1 ? 8 : 0));
    const unsigned rel_g = g_role ? (unsigned)((ghy * S + ghx) * 16) : 0xC0000000u;

    f32x4 ra[NA], rg;
    auto issue_tile = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const i32x4w_ rs_a = make_rsrc(a + (size_t)n * S * S * C, (unsigned)(S * S * C * 4));
        const i32x4w_ rs_g = make_rsrc(g + (size_t)n * S * S * 4, (unsigned)(S * S * 16));
        const unsigned tb = (ty == 0 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == tiles_x - 1 ? 8u : 0u);
        const unsigned oa = rel_a + (unsigned)((((ty * TH) * S + tx * TW) * C) * 4);
        const unsigned og = (gmask & tb) ? 0x80000000u : rel_g + (unsigned)(((ty * TH - 1) * S + tx * TW - 1) * 16);
        asm volatile("s_nop 4" :: "s"(rs_a) : "memory");
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int soff = k * a_kstride;
            asm volatile("s_nop 4" :: "s"(soff) : "memory");
            OW_BLOAD4(ra[k], oa, rs_a, soff);
        }
        asm volatile("s_nop 4" :: "s"(rs_g) : "memory");
        OW_BLOAD4(rg, og, rs_g, 0);
    };
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
    auto land_tile = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NA; ++k) asm volatile("" : "+v"(ra[k]));
        asm volatile("" : "+v"(rg));
#pragma unroll
        for (int k = 0; k < NA; ++k) *reinterpret_cast<f32x4*>(s_a + lw_a + (unsigned)(p0 + PSTEP * k) * 128) = ra[k];
        if (g_role) *reinterpret_cast<f32x4*>(s_g + tid * 4) = rg;
        if (g_inner) bsum += rg;
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // B operand of this lane: column j = nt * 32 + li = tap * 4 + co (zero from 36): g[(row + 2 - dy, col + 2 - dx)][co] of the halo tile
    const int j = nt * 32 + li;
    const bool jok = j < 36;
    const int tap = jok ? j >> 2 : 0, co = j & 3;
    const float* gp = s_g + ((2 - tap / 3) * HW + (2 - tap % 3) + half) * 4 + co + (ks * (TH / KS) * HW) * 4;
    const float* ap = reinterpret_cast<const float*>(s_a + mi * PLANE_A) + (ks * (NPXT / KS) + half) * 32 + li;

    const int t0 = t_begin + bix;
    if (t0 < t_end) {
        issue_tile(t0);
        land_tile();
        __syncthreads();
        for (int t = t0; t < t_end; t += bpx) {
            const bool has_next = t + bpx < t_end;
            if (has_next) issue_tile(t + bpx);
            w32_static_for<0, STEPS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int r = (2 * s) / TW, c = (2 * s) % TW;
                const float aval = ap[(2 * s) * 32];
                float bval = gp[(r * HW + c) * 4];
                bval = jok ? bval : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bval, acc, 0, 0, 0);
            });
            if (has_next) {
                __syncthreads();
                land_tile();
                __syncthreads();
            }
        }
    }
    // partial dW: accumulator rows = ci, columns = j -> part[(tap * C + ci) * 4 + co]
    if (jok) {
        float* pw = part + (size_t)(blockIdx.x * KS + ks) * 9 * C * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            pw[((size_t)tap * C + ci) * 4 + co] = acc[r];
        }
    }
    __syncthreads();
    f32x4* s_red = reinterpret_cast<f32x4*>(smem_ow);
    s_red[tid] = bsum;
    __syncthreads();
    if (tid == 0) {
        f32x4 t4 = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < NHALO; ++q) t4 += s_red[q];
        *reinterpret_cast<f32x4*>(part_b + (size_t)blockIdx.x * 4) = t4;
    }
#undef OW_BLOAD4
}

template <int C>
static hipError_t launch_dec_out_wgrad_f32_inst(hipStream_t st, const float* a, const float* g, float* part, float* part_b, int N, int S,
                                                int* nparts, int* nbias_parts)
{
    constexpr int KS = 2 / (C / 32);
    constexpr size_t lds = (size_t)(C / 32) * 64 * 128 + (size_t)6 * 18 * 16;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    const int ntiles = N * (S / 4) * (S / 16);
    const int per_xcd = (ntiles + 7) / 8;
    const int bpx = std::min(per_xcd, std::max(1, 4 * n_cu / 8));       // four persistent blocks per CU
    const int blocks = std::min(8 * bpx, 1024);                         // (capacity of the partial-tile buffers)
    hipLaunchKernelGGL((dec_out_wgrad_f32_kernel<C>), dim3(blocks), dim3(256), lds, st, a, g, part, part_b, S, ntiles);
    *nparts = blocks * KS;
    *nbias_parts = blocks;
    return hipGetLastError();
}

// a: NHWC [N][S][S][c], g: [N][S][S][4]; part: [nparts][9][c][4], part_b: [nbias_parts][4]
hipError_t launch_dec_out_wgrad_f32(hipStream_t st, const float* a, const float* g, float* part, float* part_b, int N, int S, int c,
                                    int* nparts, int* nbias_parts)
{
    if (S % 16 != 0 || (size_t)S * S * c * 4 >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    if (c == 64) return launch_dec_out_wgrad_f32_inst<64>(st, a, g, part, part_b, N, S, nparts, nbias_parts);
    if (c == 32) return launch_dec_out_wgrad_f32_inst<32>(st, a, g, part, part_b, N, S, nparts, nbias_parts);
    return hipErrorInvalidValue;
}
