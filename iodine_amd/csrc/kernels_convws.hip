// Weight-stationary form of the decoder's stride-1 3x3 conv C -> C (forward + bias + ELU, and the data gradient with
// transposed / flipped weights x ELU'), split-fp16 arithmetic (fp32 operands as fp16 hi + lo, three MFMAs, fp32 accumulate:
// see kernels_conv.hip).  Reference: nn.Conv2d + F.elu of MultiLayerConv, lib/modeling/iodine.py:583-592, and its autograd.
//
// Why another kernel: the LDS-tiled kernel (conv3x3_tile_f16x3_kernel) re-stages the packed weights of the layer (147 KB
// at C = 64) for every 256-pixel tile - 47 % of a block's incoming bytes, 36 of its 60 global loads and LDS stores per
// thread, half of its LDS fragment reads and two barriers per 16-channel chunk - and sits at 55 % matrix-pipe
// utilisation.  Here the weights never touch LDS:
//   * a wave OWNS 16 output channels and keeps their whole [C x 9 taps x 16] weight slice, hi and lo, in registers for the
//     life of the (persistent) block: 144 VGPRs at C = 64 (the four waves of a block hold the layer once; two blocks/CU);
//   * D[16 cout x 16 px] += W[16 cout x 32 cin] . X[32 cin x 16 px] with v_mfma_f32_16x16x32_f16: the activation fragment
//     of (input row r, column shift dx) is read from LDS ONCE and used by the three taps dy = 0..2, i.e. by three output
//     rows (9 MFMAs per two ds_read_b128): 2.7x fewer activation fragment reads per tile than one read per tap;
//   * LDS holds only the input halo of an 8 x 16 tile (10 x 18 pixels, 32 channels per chunk, hi | lo, 160-byte pixel
//     stride = conflict-free for the fragment reads), double-buffered: ONE barrier per chunk, staging of chunk q + 1 is in
//     flight under the MFMAs of chunk q;
//   * the power-of-two scale of a tile comes from per-tile max |x| values the PRODUCER of the tensor left in a side buffer
//     (tmax, 4 floats per 8 x 16 cell): no block-wide max reduction in front of the split.  Without a side buffer the block
//     reduces the max itself (one more barrier per chunk).
// Output and ELU' operand move as one float4 (4 channels) per lane and row: 16 pixels x 64 contiguous bytes per instruction.
#include "common.h"
#include "pack_bodies.h"
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- packed weights: [cout group][chunk of 32 cin][tap][hi/lo][lane][8 fp16] = the A operand of v_mfma_f32_16x16x32_f16
// (lane l: row = cout 16 cg + l % 16, k = cin 32 c + 8 (l / 16) .. + 7), pre-scaled by meta[0] (weight_scale_kernel) -------
__global__ void pack_conv_weights_ws_kernel(const float* __restrict__ src, int C, int tflip, const float* __restrict__ meta,
                                            _Float16* __restrict__ dst)
{
    const float scale = meta[0];
    const size_t total = pack_ws_total(C);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
        dst[idx] = pack_ws_element(src, C, tflip, scale, idx);
}

__global__ __launch_bounds__(1024) void weight_scale_ws_kernel(const float* __restrict__ w, int n, float* __restrict__ meta)
{
    __shared__ float s_red[16];
    weight_scale_block(w, n, meta, s_red);
}

hipError_t launch_pack_conv_weights_ws(hipStream_t st, const float* src, int C, int tflip, float* meta, void* dst)
{
    hipLaunchKernelGGL(weight_scale_ws_kernel, dim3(1), dim3(1024), 0, st, src, C * C * 9, meta);
    const size_t total = (size_t)(C / 16) * (C / 32) * 9 * 2 * 64 * 8;
    hipLaunchKernelGGL(pack_conv_weights_ws_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, C, tflip,
                       meta, (_Float16*)dst);
    return hipGetLastError();
}

// ---- per-cell max |x| of an NHWC tensor (cells of 8 rows x 16 columns, 4 identical floats per cell): the side buffer for
// tensors whose producer does not emit it (op-level tests, first use of a tensor) ------------------------------------------
__global__ __launch_bounds__(256) void cell_max_kernel(const float* __restrict__ x, float* __restrict__ tmax, int S, int C)
{
    const int cells_x = S / 16, cells_y = S / 8;
    const int cell = blockIdx.x;
    const int cx = cell % cells_x, cy = (cell / cells_x) % cells_y, n = cell / (cells_x * cells_y);
    const float4* base = reinterpret_cast<const float4*>(x + (size_t)n * S * S * C);
    const int c4 = C / 4;
    float m = 0.f;
    for (int i = threadIdx.x; i < 128 * c4; i += 256) {
        const int px = i / c4, q = i % c4;
        const float4 v = base[((size_t)(cy * 8 + px / 16) * S + cx * 16 + px % 16) * c4 + q];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    __shared__ float s_red[4];
    m = wave_max_f32(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 4) tmax[(size_t)cell * 4 + threadIdx.x] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

hipError_t launch_cell_max(hipStream_t st, const float* x, float* tmax, int N, int S, int C)
{
    hipLaunchKernelGGL(cell_max_kernel, dim3(N * (S / 16) * (S / 8)), dim3(256), 0, st, x, tmax, S, C);
    return hipGetLastError();
}

#ifdef IODINE_TILE_PROF
__device__ unsigned g_ws_prof[TP_MAXBLK * 8];
__device__ unsigned long long g_ws_ts[TP_MAXBLK * 16];
__device__ unsigned long long g_ws_se[TP_MAXBLK * 2];       // [block]: s_memtime at the block's start / end
#endif

template <int I, int N, typename F>
IOD_DEVINL void ws_static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ws_static_for<I + 1, N>(f);
    }
}

// power-of-two scale that puts max |x| = mx into [2^12, 2^13) (a pure function of mx: results must not depend on which
// block processed which tile before)
IOD_DEVINL float fresh_scale(float mx)
{
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
    int se = 12 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return __uint_as_float((unsigned)(127 + se) << 23);
}

// sum over the pixel lanes of the whole-pixel layout (lane = pl * SEGS + seg): all lanes with the same seg, result in every
// lane.  gfx950 lane swaps / DPP, no LDS: v_permlane32_swap (halves), v_permlane16_swap (odd / even rows of 16), row_ror:8.
template <int SEGS>
IOD_DEVINL float pixel_lane_sum(float v)
{
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if constexpr (SEGS == 8) v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));
    return v;
}

// LDS / occupancy: 160-byte staged pixels, two persistent blocks per CU.  Round 5 tried THREE blocks per CU at C = 32 (144-byte pixels:
// 3 x 52 KB of LDS, 165 VGPRs): the cfg2 step did not move (6.57 -> 6.54 ms incl. the pixel-kernel gains of the same build).  The phase
// counters say why: the three waves of a SIMD run their MFMA phases in lockstep (4.9 k ticks per tile for 3 x 1.7 k cycles of MFMA:
// the pipe is saturated while they are in it) and their epilogues too; per launch only 8 - 12 tiles per block follow a ~8 us prologue
// (weights -> registers, first halo: two exposed HBM round trips) - at 64 x 64 images the launch is latency-, not occupancy-bound.
constexpr int ws_pxb(int c) { return 160; }
constexpr int ws_bpc(int c, int epi) { return 2; }

template <int C, int EPI, bool F32 = false>
__global__ __launch_bounds__(256, ws_bpc(C, EPI))
void conv3x3_ws_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                             const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                             const float* __restrict__ tmax_in, float* __restrict__ tmax_out, int S, int lgS, int ntiles, int rev)
{
    constexpr int NCG = C / 16;                  // cout groups = waves along the channel axis
    constexpr int NPG = 4 / NCG;                 // pixel groups (C = 32: two waves share a cout group, 4 rows each)
    constexpr int RW = 8 / NPG;                  // output rows per wave
    constexpr int NCHUNK = C / 32;               // K chunks of 32 input channels
    constexpr int HC = 18, HR = 10, NPX = HC * HR;
    constexpr int PXB = ws_pxb(C);               // bytes per staged pixel: 64 hi | 64 lo | 32 (C = 32: 16) pad (conflict-free ds_read_b128);
                                                 // exact-fp32 form (F32): 32 channels x 4 bytes | 32 pad - the same geometry
    constexpr int EPS = C * 4 + 32;              // bytes per pixel of the transposed output tile (epilogue)
    constexpr int BUFB = ((NPX + 1) * PXB > 128 * EPS ? (NPX + 1) * PXB : 128 * EPS);    // one LDS buffer (input halo / output tile)
    constexpr int NIN = (NPX * 8 + 255) / 256;   // float4 loads per thread per chunk (6)
    constexpr int NS = (RW + 2) * 3;             // fragment steps per chunk: halo rows x column shifts
    constexpr bool XPOSE = true;                 // epilogue through LDS: whole-pixel (1 KB contiguous) loads / stores (every form)
    constexpr bool ROWS = EPI == EPI_L0ROWS || EPI == EPI_L0ROWSX;        // row-sum forms: nothing stored per pixel
    constexpr int NQ = EPI == EPI_L0ROWSX ? 4 : 3;                        // sums per row and tile
    constexpr bool GRADF = EPI == EPI_MUL_ELUGRAD || ROWS;                // data-gradient forms: x ELU'(aux)
    constexpr int SEGS = C / 4, PPI = 64 / SEGS, NEP = 32 / PPI;      // whole-pixel layout: float4 segments, pixels / instruction
    static_assert(C == 64 || C == 32, "channel counts of the shipped decoders");
    static_assert(EPI == EPI_BIAS_ELU || EPI == EPI_MUL_ELUGRAD || ROWS, "epilogue");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

    using std::integral_constant;
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wv % NCG, pg = wv / NCG;
    const int lpx = lane & 15, lkb = lane >> 4;

    auto make_rsrc = [&](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)base;
        i32x4_ r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
#define WS_BLOAD4(dst, voff, rsrc, soff) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory")
// cache policy of the output stores (experiment hook): 0 default, 1 nt, 2 sc1, 3 sc0 sc1
#ifndef WS_STORE_POLICY
#define WS_STORE_POLICY 0
#endif
#if WS_STORE_POLICY == 1
#define WS_STORE_MOD " nt"
#elif WS_STORE_POLICY == 2
#define WS_STORE_MOD " sc1"
#elif WS_STORE_POLICY == 3
#define WS_STORE_MOD " sc0 sc1"
#else
#define WS_STORE_MOD ""
#endif
#define WS_BSTORE4(src, voff, rsrc, soff) \
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" WS_STORE_MOD "\n\ts_nop 1" :: "v"(src), "v"(voff), "s"(rsrc), "s"(soff) : "memory")
#define WS_SGPR_SETTLE(rsrc, soff) asm volatile("s_nop 4" :: "s"(rsrc), "s"(soff) : "memory")

    // ---- this wave's weight slice -> registers (once per block) ----
    // (F32: the same 16 bytes per lane are 4 fp32 weights - row = cout 16 cg + l % 16, cin 32 c + 4 (l / 16) .. + 3 in wh and
    // + 16 in wl: element j is the A operand of the j-th v_mfma_f32_16x16x4_f32 over the fragment's 16 bytes)
    using wreg_t = std::conditional_t<F32, f32x4, f16x8>;
    wreg_t wh[NCHUNK][9], wl[NCHUNK][9];
    // (round 5: filled in the prologue BEHIND the first stage's input loads, so that the block pays one exposed memory round trip at its
    // start instead of two - at 64 x 64 images a block has only 8 - 12 tiles to amortise its prologue over)
    auto load_weights = [&]() {
        const uint4* wp = wpk + (size_t)cg * NCHUNK * 9 * 2 * 64 + lane;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint4 h = wp[((c * 9 + t) * 2 + 0) * 64], l = wp[((c * 9 + t) * 2 + 1) * 64];
                __builtin_memcpy(&wh[c][t], &h, 16);
                __builtin_memcpy(&wl[c][t], &l, 16);
            }
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(wh[c][t]), "+v"(wl[c][t]));      // opaque: stay in registers
    };
    float inv_w = 1.f;
    if constexpr (!F32) inv_w = wmeta[1];
    // (training row-sum form) step of torch.linspace(-1, 1, S), wave-uniform: computed once, kept in a scalar register
    const float xstep_u = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(__fdiv_rn(2.f, (float)(S - 1)))));
    (void)xstep_u;

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_b;
    // staging: float4 k of this thread = halo pixel (tid >> 3) + 32 k, channel quad tid & 7 (idle lanes of the last one write
    // the dump slot); fragment reads: pixel (row pg*RW, column lpx), 16-byte k block lkb
    const unsigned lw0 = (unsigned)((tid >> 3) * PXB + (tid & 7) * (F32 ? 16 : 8));
    const bool last_idle = tid + (NIN - 1) * 256 >= NPX * 8;
    const unsigned fr_base = lds_base + (unsigned)(((pg * RW) * HC + lpx) * PXB + lkb * 16);

    // (S is a power of two here - the launcher routes other sizes to the LDS-tiled kernel - so tile coordinates are shifts)
    const int tiles_x = S >> 4, tiles_y = S >> 3, lg_tx = lgS - 4, lg_tpi = 2 * lgS - 7;
    // XCD-aware persistent schedule: block b runs on XCD b % 8 (observed; speed only).  Every XCD gets one contiguous eighth
    // of the tile list, so that the blocks sharing an L2 work on neighbouring tiles / the same slot-images.
    const int nblk = gridDim.x;
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = (nblk + 7) >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_begin = xcd * per_xcd, t_end = min(ntiles, t_begin + per_xcd);

    f32x4 rin[NIN];
    auto tile_coords = [&](int t, int& n, int& ty, int& tx) {
        const int tt = rev ? ntiles - 1 - t : t;
        n = tt >> lg_tpi;
        ty = (tt >> lg_tx) & (tiles_y - 1); tx = tt & (tiles_x - 1);
    };
    // side buffer prefetch: lane i < 36 fetches float i & 3 of cell (i >> 2) of the 3 x 3 neighbourhood (clamped)
    float tmv = 0.f;                                     // this lane's prefetched cell max of the tile whose chunk 0 is in flight
    auto issue_tmax = [&](int t) {
        int n, ty, tx;
        tile_coords(t, n, ty, tx);
        const int cell = min(lane >> 2, 8);
        const int cy = min(max(ty + cell / 3 - 1, 0), tiles_y - 1), cx = min(max(tx + cell % 3 - 1, 0), tiles_x - 1);
        const float* p = tmax_in + ((((size_t)n << lg_tpi) + (cy << lg_tx) + cx) << 2) + (lane & 3);
        asm volatile("global_load_dword %0, %1, off" : "=v"(tmv) : "v"(p) : "memory");
    };
    auto issue_loads = [&](int t, int chunk) {
        int n, ty, tx;
        tile_coords(t, n, ty, tx);
        const i32x4_ rsrc = make_rsrc(in + (size_t)n * S * S * C, (unsigned)(S * S * C * 4));
        unsigned goff[NIN];
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int px = (tid >> 3) + 32 * k;
            const int hy = (px * 3641) >> 16, hx = px - hy * HC;             // px / 18 for px < 2048
            const int gy = ty * 8 - 1 + hy, gx = tx * 16 - 1 + hx;
            const bool ok = (k < NIN - 1 || px < NPX) && (unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S;
            goff[k] = ok ? (unsigned)(((((gy << lgS) + gx) * C) << 2) + ((tid & 7) << 4)) : 0x80000000u;
        }
        const int soff = chunk * 128;
        WS_SGPR_SETTLE(rsrc, soff);
#pragma unroll
        for (int k = 0; k < NIN; ++k) WS_BLOAD4(rin[k], goff[k], rsrc, soff);
    };
    // stage s of this block = (tile t0 + (s / NCHUNK) * bpx, chunk s % NCHUNK)
    const int t0 = t_begin + bix;
    const int nstage = t0 < t_end ? ((t_end - t0 + bpx - 1) / bpx) * NCHUNK : 0;
    auto issue_stage = [&](int s) {
        const int t = t0 + (s / NCHUNK) * bpx, c = s % NCHUNK;
        issue_loads(t, c);
        if constexpr (!F32) { if (c == 0) issue_tmax(t); }
    };
    auto vm_wait = [&](auto nc) {
        constexpr int nleft = decltype(nc)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(nleft) : "memory");
#pragma unroll
        for (int k = 0; k < NIN; ++k) asm volatile("" : "+v"(rin[k]));
        asm volatile("" : "+v"(tmv));
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4 acc[RW];

    // float4 k of the staged registers -> fp16 (hi, lo) at `scale` -> LDS buffer `buf`
    auto convert_k = [&](auto kc, int buf, float scale) {
        constexpr int k = decltype(kc)::value;
        unsigned char* sb = smem_b + buf * BUFB;
        f32x4 v = rin[k];
        if constexpr (F32) {                                 // exact fp32: the float4 goes to LDS as it is
            const unsigned o32 = (k == NIN - 1 && last_idle) ? (unsigned)(NPX * PXB) : lw0 + (unsigned)(k * 32 * PXB);
            *reinterpret_cast<f32x4*>(sb + o32) = v;
        } else {
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        typedef __fp16 h2 __attribute__((ext_vector_type(2)));
        const h2 h01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), h23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);      // (rounding toward zero = the 11 leading bits)
        const h2 l01 = __builtin_amdgcn_cvt_pkrtz(v.x - (float)h01.x, v.y - (float)h01.y), l23 = __builtin_amdgcn_cvt_pkrtz(v.z - (float)h23.x, v.w - (float)h23.y);   // v_fma_mix_f32
        uint2 hi, lo;
        __builtin_memcpy(&hi.x, &h01, 4); __builtin_memcpy(&hi.y, &h23, 4);
        __builtin_memcpy(&lo.x, &l01, 4); __builtin_memcpy(&lo.y, &l23, 4);
        const unsigned o = (k == NIN - 1 && last_idle) ? (unsigned)(NPX * PXB) : lw0 + (unsigned)(k * 32 * PXB);
        *reinterpret_cast<uint2*>(sb + o) = hi;
        *reinterpret_cast<uint2*>(sb + o + 64) = lo;
        }
    };

    // stores (and the tile-max store) an epilogue leaves in flight: YOUNGER than the input loads of the stage after it
    // (row-sum forms: per wave and tile 2 rows x {left, interior, right [, weighted]} float4 stores by the lanes of pixel lane 0)
    constexpr int NST = ROWS ? 2 * NQ : NEP + (F32 ? 0 : 1);

#define WS_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    struct Frag { wreg_t h, l; };
    // All taps of one 32-channel chunk from LDS buffer `buf`: (RW + 2) halo rows x 3 column shifts, each fragment pair feeding
    // the taps dy = 0..2 = output rows hr - dy.  The staging of the NEXT stage rides inside: its loads (in flight since the
    // previous chunk) are waited for after step HOOK0, split + written to the other LDS buffer one float4 per step, and the
    // loads of the stage after that are issued - all between MFMAs, none of it in front of the barrier.
    constexpr int HOOK0 = 3;
    auto compute = [&](auto cc, int buf, bool has_next, bool next_after_epi, float next_scale_same, float& next_scale, bool has_next2,
                       int next2_stage) {
        constexpr int c = decltype(cc)::value;
        const unsigned base = fr_base + (unsigned)(buf * BUFB);
        Frag f[2];
        auto LOADF = [](auto sc, Frag& fr, unsigned b) {
            constexpr int s = decltype(sc)::value;
            constexpr int hr = s / 3, dx = s % 3;
            constexpr int off = (hr * HC + dx) * PXB;
            WS_DSR128(fr.h, b, off);
            WS_DSR128(fr.l, b, off + 64);
        };
        auto MMA = [&](auto sc, const Frag& fr) {
            constexpr int s = decltype(sc)::value;
            constexpr int hr = s / 3, dx = s % 3;
            if constexpr (F32) {
                // exact fp32: 8 k-steps of 4 channels (v_mfma_f32_16x16x4_f32: 32 cycles each, 40 cycles dependent latency - the
                // dy loop inside, so that consecutive MFMAs go to different accumulators)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int y = hr - dy;
                        if (y >= 0 && y < RW) acc[y] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[c][dy * 3 + dx][j], fr.h[j], acc[y], 0, 0, 0);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int y = hr - dy;
                        if (y >= 0 && y < RW) acc[y] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[c][dy * 3 + dx][j], fr.l[j], acc[y], 0, 0, 0);
                    }
            } else {
            // three passes, accumulators interleaved (consecutive MFMAs never share an accumulator)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int y = hr - dy;
                if (y >= 0 && y < RW) acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c][dy * 3 + dx], fr.l, acc[y], 0, 0, 0);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int y = hr - dy;
                if (y >= 0 && y < RW) acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c][dy * 3 + dx], fr.h, acc[y], 0, 0, 0);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int y = hr - dy;
                if (y >= 0 && y < RW) acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c][dy * 3 + dx], fr.h, acc[y], 0, 0, 0);
            }
            }
        };
        LOADF(integral_constant<int, 0>{}, f[0], base);
        auto step = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s + 1 < NS) {
                LOADF(integral_constant<int, (s + 1 < NS ? s + 1 : 0)>{}, f[(s + 1) & 1], base);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("" : "+v"(f[s & 1].h), "+v"(f[s & 1].l));
            __builtin_amdgcn_sched_barrier(0);
#ifndef WS_ABL_NOMFMA
            MMA(sc, f[s & 1]);
#endif
            // ---- staging hooks (block-uniform conditions) ----
            if constexpr (s == HOOK0) {
                if (has_next) {
                    if (next_after_epi) vm_wait(integral_constant<int, NST>{});
                    else vm_wait(integral_constant<int, 0>{});
                    // the next stage opens a tile: its scale comes from the side buffer values fetched with its loads
                    if constexpr (!F32) next_scale = c + 1 < NCHUNK ? next_scale_same : fresh_scale(wave_max_f32(lane < 36 ? tmv : 0.f));
                }
            }
            if constexpr (s > HOOK0 && s <= HOOK0 + NIN) {
                if (has_next) convert_k(integral_constant<int, s - HOOK0 - 1>{}, buf ^ 1, next_scale);
            }
            if constexpr (s == HOOK0 + NIN + 1) {
                if (has_next2) issue_stage(next2_stage);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        ws_static_for<0, NS>(step);                          // straight-line code, literal LDS offsets
    };
#undef WS_DSR128

    if (nstage == 0) return;
#ifdef IODINE_TILE_PROF
    if (tid == 0 && blockIdx.x < TP_MAXBLK) g_ws_se[blockIdx.x * 2] = __builtin_amdgcn_s_memtime();
#endif
#ifdef WS_ROLEPRIO
    // Two persistent blocks share a CU (one wave of each per SIMD).  Left alone they fall into LOCKSTEP - whoever lags gets the
    // matrix pipe to itself and catches up - so both sit in their epilogues at the same time and the pipe idles.  A static
    // priority difference breaks the symmetry: the favoured wave runs its MFMA phases unimpeded, the other one fills the pipe
    // whenever the favoured one is in an epilogue or stalls.  Role = parity of the hardware wave slot (the two waves of a SIMD
    // sit in different slots); if the guess fails both keep the same priority and nothing is lost.
    {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (hwid & 1u) __builtin_amdgcn_s_setprio(WS_ROLEPRIO);
    }
#endif
#ifndef WS_ALTPRIO
#define WS_ALTPRIO 2
#endif
#ifndef WS_ALT_SH32
#define WS_ALT_SH32 18                           // favoured-role period of the clock-based alternation: 2^18 ticks (fp32 form), 2^16 (split-fp16)
#endif
#ifndef WS_ALT_SH16
#define WS_ALT_SH16 16
#endif
#if !defined(WS_ALTTIME) && !defined(WS_ALTTILES)
#define WS_ALTTIME 1                             // (WS_ALTTILES: the tile-count form of rounds 2 - 4, kept for A/B builds)
#endif
#if WS_ALTPRIO > 0
    unsigned role;
    { unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); role = hwid & 1u; }
#endif
    TP_DECL;
    // ---- prologue: stage 0 into buffer 0, stage 1 in flight ----
    issue_stage(0);
    load_weights();                                          // (its own wait covers the stage-0 loads too: one round trip for both)
    vm_wait(integral_constant<int, 0>{});
    float cur_scale = 1.f;
    if constexpr (!F32) cur_scale = fresh_scale(wave_max_f32(lane < 36 ? tmv : 0.f));
    ws_static_for<0, NIN>([&](auto kc) { convert_k(kc, 0, cur_scale); });
    if (nstage > 1) issue_stage(1);
    bool first = true;
    float next_scale = cur_scale;
#ifdef WS_PRIO
    __builtin_amdgcn_s_setprio(WS_PRIO);
#endif
    for (int s0 = 0; s0 < nstage; s0 += NCHUNK) {
#if WS_ALTPRIO > 0
        // the older of the two waves of a SIMD wins every arbitration and runs ~1.5x faster than its partner (56 vs 37 tiles in
        // the same time, then a long tail alone): swap the favoured role every WS_ALTPRIO tiles so that both progress alike
#ifdef WS_ALTTIME
        // (round 5) ... by the CLOCK, not by the block's own tile count: counted in tiles the two blocks of a CU drift apart until both hold
        // the same priority at the same time (then the older wave wins every arbitration again): the block in wave slot 1 finished 354 k
        // ticks = 4.5 tiles of 56 after its partner and ran that tail alone.  s_memtime is one counter per XCD: exactly one of the two is
        // favoured at any time, for ~2 - 3 tiles
        if ((((unsigned)(__builtin_amdgcn_s_memtime() >> (F32 ? WS_ALT_SH32 : WS_ALT_SH16))) ^ role) & 1u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#else
        if ((((unsigned)(s0 / NCHUNK) / WS_ALTPRIO) ^ role) & 1u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
#endif
        const int t = t0 + (s0 / NCHUNK) * bpx;
        int n, ty, tx;
        tile_coords(t, n, ty, tx);
#pragma unroll
        for (int y = 0; y < RW; ++y) acc[y] = f32x4{0.f, 0.f, 0.f, 0.f};
        TP_STAMP(0);                                         // [0] tile bookkeeping
#ifdef IODINE_TILE_PROF
        { const int ti = s0 / NCHUNK - 20; if (tid == 0 && ti >= 0 && ti < 8 && blockIdx.x < TP_MAXBLK) g_ws_ts[blockIdx.x * 16 + 2 * ti] = __builtin_amdgcn_s_memtime(); }
#endif
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            const int s = s0 + c;
            __syncthreads();                                 // stage s is complete in buffer s & 1; buffer (s + 1) & 1 is free
            TP_STAMP(1);                                     // [1] barrier
#ifdef WS_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            // the stage after this one was issued one chunk ago; its loads are older than the stores of the epilogue that
            // ran in between, if one did (chunk 0 of every tile but the block's first)
            const bool after_epi = c == 0 && !first;
            if (c == 0) compute(integral_constant<int, 0>{}, s & 1, s + 1 < nstage, after_epi, cur_scale, next_scale, s + 2 < nstage, s + 2);
            else compute(integral_constant<int, (NCHUNK > 1 ? 1 : 0)>{}, s & 1, s + 1 < nstage, after_epi, cur_scale, next_scale, s + 2 < nstage, s + 2);
#ifdef WS_PRIO
            __builtin_amdgcn_s_setprio(WS_PRIO);
#endif
            TP_STAMP(2);                                     // [2] fragment reads + MFMAs (+ staging hooks) of the chunk
        }
        first = false;
#ifdef IODINE_TILE_PROF
        { const int ti = s0 / NCHUNK - 20; if (tid == 0 && ti >= 0 && ti < 8 && blockIdx.x < TP_MAXBLK) g_ws_ts[blockIdx.x * 16 + 2 * ti + 1] = __builtin_amdgcn_s_memtime(); }
#endif
        // ---- epilogue ----
        const float inv = inv_w / cur_scale;
        cur_scale = next_scale;                              // (the scale of the tile whose chunk 0 was just staged)
        const i32x4_ rsrc_out = make_rsrc(out + (size_t)n * S * S * C, (unsigned)(S * S * C * 4));
        if constexpr (XPOSE) {
            // Through LDS, so that global traffic is whole pixels (1 KB contiguous per instruction; the direct form - 16 pixels
            // x 64 bytes per instruction - cost 7 % of the kernel in partial-line stores).  The tile goes into the buffer the last
            // chunk was read from: every wave must be done reading it (barrier), and the next stage's hooks write it again only
            // behind the next stage barrier.
            unsigned char* sb = smem_b + ((s0 + NCHUNK - 1) & 1) * BUFB;
            const int seg = lane % SEGS, pl = lane / SEGS;
            // wave wv moves tile rows 2 wv, 2 wv + 1: instruction j = pixels j * PPI .. of those 32
            const unsigned vbase = (unsigned)(((((ty * 8 + 2 * wv) << lgS) + tx * 16 + pl) * C + seg * 4) * 4);
            f32x4 ax[GRADF ? NEP : 1];
            if constexpr (GRADF) {
                // the ELU' operand (whole pixels, like the stores) is requested before the barriers: in the training step it
                // comes from HBM, not from a cache
                // (first half now - the fragment registers are free -, second half once the accumulators are in LDS)
                const i32x4_ rsrc_aux = make_rsrc(aux + (size_t)n * S * S * C, (unsigned)(S * S * C * 4));
#pragma unroll
                for (int j = 0; j < NEP / 2; ++j) {
                    const int soff = ((((j * PPI) / 16) << lgS) + (j * PPI) % 16) * C * 4;
                    WS_SGPR_SETTLE(rsrc_aux, soff);
                    WS_BLOAD4(ax[j], vbase, rsrc_aux, soff);
                }
            }
            __syncthreads();
            TP_STAMP(3);                                     // [3] barrier (tile buffer free)
#pragma unroll
            for (int y = 0; y < RW; ++y)
                *reinterpret_cast<f32x4*>(sb + ((pg * RW + y) * 16 + lpx) * EPS + (16 * cg + 4 * lkb) * 4) = acc[y] * inv;
            if constexpr (GRADF) {
                const i32x4_ rsrc_aux = make_rsrc(aux + (size_t)n * S * S * C, (unsigned)(S * S * C * 4));
#pragma unroll
                for (int j = NEP / 2; j < NEP; ++j) {
                    const int soff = ((((j * PPI) / 16) << lgS) + (j * PPI) % 16) * C * 4;
                    WS_SGPR_SETTLE(rsrc_aux, soff);
                    WS_BLOAD4(ax[j], vbase, rsrc_aux, soff);
                }
            }
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == EPI_BIAS_ELU) {
                const float4 tb = *reinterpret_cast<const float4*>(bias + seg * 4);
                b4 = f32x4{tb.x, tb.y, tb.z, tb.w};
            }
            __syncthreads();
            TP_STAMP(4);                                     // [4] tile -> LDS + barrier
            float vmax = 0.f;
            const unsigned char* sr = sb + ((2 * wv) * 16 + pl) * EPS + seg * 16;
            // all LDS reads first (the asm stores below are memory barriers for the compiler: interleaved with them every
            // read would be issued, and waited for, on its own)
            // (data-gradient form: in two batches - the ELU' operand occupies NEP float4s of its own)
            constexpr int PB = GRADF ? NEP / 2 : NEP;
            // (training row-sum form) x-coordinate-weighted row sum: a fourth running sum next to the others does not fit the
            // register budget while the first half of the ELU' operand is live (3 - 11 spilled loop invariants, each reload a
            // vmcnt(0) inside the load-issue hook: +30 % kernel time), so the wave's FIRST tile row leaves its finished values
            // in LDS and is weighted in a short second pass; the second row is weighted on the fly.
            constexpr int XJ0 = EPI == EPI_L0ROWSX ? NEP / 2 : NEP;    // instructions j >= XJ0 accumulate the weighted sum directly
            f32x4 rxw = f32x4{0.f, 0.f, 0.f, 0.f};
            (void)rxw;
            // torch.linspace(-1, 1, S)[x] as iodine_linspace_host builds it (separately rounded multiply and add)
            auto lin_at = [&](int x) {
                return x < (S >> 1) ? __fadd_rn(-1.f, __fmul_rn(xstep_u, (float)x)) : __fsub_rn(1.f, __fmul_rn(xstep_u, (float)(S - 1 - x)));
            };
            auto xw_row_done = [&](int row) {                          // lane sums + store of one row's weighted sum
#pragma unroll
                for (int e = 0; e < 4; ++e) rxw[e] = pixel_lane_sum<SEGS>(rxw[e]);
                if (pl == 0) {
                    float* rpx = out + ((((size_t)n * S + ty * 8 + 2 * wv + row) * tiles_x + tx) * NQ) * C + seg * 4;
                    *reinterpret_cast<f32x4*>(rpx + 3 * C) = rxw;
                }
            };
            (void)lin_at; (void)xw_row_done;
            f32x4 pv[PB];
            f32x4 rtot = f32x4{0.f, 0.f, 0.f, 0.f};                    // (row-sum forms) running interior sum of the current tile row
            (void)rtot;
            ws_static_for<0, NEP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j % PB == 0) {
#pragma unroll
                    for (int i = 0; i < PB; ++i) pv[i] = *reinterpret_cast<const f32x4*>(sr + (j + i) * PPI * EPS);
                }
                f32x4 v = pv[j % PB];
                if constexpr (EPI == EPI_BIAS_ELU) {
                    // (the compare form of ELU: a max / min formulation would turn NaN inputs into 0 - the reference propagates them)
                    v = f32x4{elu1_fast(v.x + b4.x), elu1_fast(v.y + b4.y), elu1_fast(v.z + b4.z), elu1_fast(v.w + b4.w)};
                } else {
                    // aux loads return in order; behind ax[j]: NEP - 1 - j younger aux loads + (storing form) the j stores already issued
                    f32x4& axj = ax[j];
                    // (row-sum forms: + the row-sum stores issued so far - 3 per finished row, the left-border one of this row;
                    // vmcnt counts in issue order, so younger stores need not be waited for)
                    constexpr int IPRW = 16 / PPI;
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ROWS ? NEP - 1 - j + 3 * (j / IPRW) + (j % IPRW ? 1 : 0) : NEP - 1) : "memory");
                    asm volatile("" : "+v"(axj));
                    v.x *= elu1_grad_from_out(axj.x); v.y *= elu1_grad_from_out(axj.y);
                    v.z *= elu1_grad_from_out(axj.z); v.w *= elu1_grad_from_out(axj.w);
                }
                if constexpr (EPI == EPI_L0ROWSX) {
                    constexpr int IPRX = 16 / PPI;
                    if constexpr (j < XJ0) {          // first row: the finished value goes back to this lane's own LDS slot
#ifndef WS_ABL_X_NOWB
                        *reinterpret_cast<f32x4*>(const_cast<unsigned char*>(sr) + j * PPI * EPS) = v;
#endif
                    } else {
#ifndef WS_ABL_X_NOINLINE
                        const float xwj = lin_at(tx * 16 + pl + (j % IPRX) * PPI);
                        if constexpr (j % IPRX == 0) rxw = v * xwj;
                        else rxw += v * xwj;
                        if constexpr (j % IPRX == IPRX - 1) xw_row_done(j / IPRX);
#endif
                    }
                }
                if constexpr (ROWS) {
                    // Row-sum forms (layer 1): d(pre-activation 0) is not stored but reduced to per-row left-border / interior /
                    // right-border column sums rows_p[n][gy][tx][NQ][C] (the input of l0_reduce_cls_tiles); the training form
                    // adds sum_x lin[x] * v over all columns of the tile (coordinate-channel gradients of the broadcast layer).
                    // Instruction j = pixels (j % IPR) * PPI + pl of tile row j / IPR; the sum over the PPI pixel lanes is lane
                    // swaps in the VALU (lane = pl * SEGS + seg; __shfl_xor would go through the LDS crossbar); border pixels
                    // exist only in the first / last tile column.
                    constexpr int IPR = 16 / PPI;
                    const bool lt = tx == 0, rt = tx == tiles_x - 1;
                    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                    float* rp = out + ((((size_t)n * S + ty * 8 + 2 * wv + j / IPR) * tiles_x + tx) * NQ) * C + seg * 4;
                    if constexpr (j % IPR == 0) {
                        // image column 0 is pixel lane 0 of the row's first instruction: stored at once, kept out of the interior sum
                        const bool isl = lt && pl == 0;
                        if (pl == 0) *reinterpret_cast<f32x4*>(rp) = isl ? v : zero;
                        rtot = isl ? zero : v;
                    } else if constexpr (j % IPR == IPR - 1) {
                        const bool isr = rt && pl == PPI - 1;
                        f32x4 Rb = isr ? v : zero, tot = rtot + (isr ? zero : v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) tot[e] = pixel_lane_sum<SEGS>(tot[e]);
                        if (rt) {                                     // block-uniform
#pragma unroll
                            for (int e = 0; e < 4; ++e) Rb[e] = pixel_lane_sum<SEGS>(Rb[e]);
                        }
                        if (pl == 0) {
                            *reinterpret_cast<f32x4*>(rp + C) = tot;
                            *reinterpret_cast<f32x4*>(rp + 2 * C) = Rb;
                        }
                    } else {
                        rtot += v;
                    }
                } else {
                    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                    const unsigned vb = vbase; const i32x4_ ro = rsrc_out;
                    const int soff = ((((j * PPI) / 16) << lgS) + (j * PPI) % 16) * C * 4;
#ifdef WS_ABL_NOSTORE
                    asm volatile("" :: "v"(v), "v"(vb), "s"(ro), "s"(soff) : "memory");
#else
                    WS_BSTORE4(v, vb, ro, soff);
#endif
                }
            });
#ifndef WS_ABL_X_NOPASSB
            if constexpr (EPI == EPI_L0ROWSX) {
                // second pass over the wave's first tile row (values left in LDS above; the ELU' operand is dead by now)
                constexpr int IPRX = 16 / PPI;
                static_assert(XJ0 == IPRX && XJ0 <= PB, "one tile row, one LDS read batch");
#pragma unroll
                for (int i = 0; i < XJ0; ++i) pv[i] = *reinterpret_cast<const f32x4*>(sr + i * PPI * EPS);
#pragma unroll
                for (int i = 0; i < XJ0; ++i) {
                    const float xwj = lin_at(tx * 16 + pl + i * PPI);
                    if (i == 0) rxw = pv[i] * xwj;
                    else rxw += pv[i] * xwj;
                }
                xw_row_done(0);
            }
#endif
            if constexpr (!ROWS && !F32) {
                // the wave's share of the cell max of this OUTPUT tile (side buffer for the consumer of `out`)
                vmax = wave_max_f32(vmax);
                const int tt = rev ? ntiles - 1 - t : t;
                float* tp = tmax_out + (size_t)tt * 4 + wv;
                asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" :: "v"(tp), "v"(vmax) : "memory");
            }
            TP_STAMP(5);                                     // [5] whole-pixel epilogue
        }
    }
    TP_FLUSH(g_ws_prof);
#ifdef IODINE_TILE_PROF
    if (tid == 0 && blockIdx.x < TP_MAXBLK) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_ws_prof[blockIdx.x * 8 + 6] = hwid; g_ws_prof[blockIdx.x * 8 + 7] = xcc;
        g_ws_se[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
    }
#endif
#undef WS_BLOAD4
#undef WS_BSTORE4
#undef WS_SGPR_SETTLE
}

template <int C, int EPI, bool F32 = false>
static hipError_t launch_ws_inst(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias,
                                 const float* aux, float* out, const float* tmax_in, float* tmax_out, int N, int S, int rev)
{
    constexpr size_t buf_in = (size_t)(18 * 10 + 1) * ws_pxb(C), buf_out = (size_t)128 * (C * 4 + 32);
    constexpr size_t lds = 2 * (buf_in > buf_out ? buf_in : buf_out);
    static std::atomic<unsigned> attr_devs{0};                             // devices this instance is configured on
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_ws_f16x3_kernel<C, EPI, F32>, (int)lds, attr_devs); e != hipSuccess) return e;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    int lgS = 0;
    while ((1 << lgS) < S) ++lgS;
    const int ntiles = N * (S / 16) * (S / 8);
    const int per_xcd = (ntiles + 7) / 8;
    int bpc = ws_bpc(C, EPI);                                                   // persistent blocks per CU (see ws_bpc)
#ifdef WS_TUNE_ENV
    if (const char* e = getenv("IODINE_WS_BPC")) bpc = atoi(e);
#endif
    const int bpx = std::min(per_xcd, std::max(1, bpc * n_cu / 8));
    hipLaunchKernelGGL((conv3x3_ws_f16x3_kernel<C, EPI, F32>), dim3(8 * bpx), dim3(256), lds, st, in,
                       reinterpret_cast<const uint4*>(wpk), wmeta, bias, aux, out, tmax_in, tmax_out, S, lgS, ntiles, rev);
#ifdef IODINE_TILE_PROF
    {
        const int nb = std::min(8 * bpx, TP_MAXBLK);
        std::vector<unsigned> hp((size_t)nb * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_ws_prof), hp.size() * sizeof(unsigned));
        static const char* names[8] = {"bookkeeping", "stage-barrier", "taps+staging-hooks", "epi-barrier", "tile->lds+barrier", "epilogue",
                                       "-", "-"};
        double sum[8] = {0}, tot = 0;
        for (int b2 = 0; b2 < nb; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[(size_t)b2 * 8 + i];
        for (int i = 0; i < 8; ++i) tot += sum[i] / nb;
        fprintf(stderr, "[ws prof] memtime ticks per block (wave 0, %d tiles), total %.0f:", (ntiles + 8 * bpx - 1) / (8 * bpx), tot);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f |", names[i], sum[i] / nb);
        fprintf(stderr, "\n");
        if (getenv("IODINE_WS_TS")) {
            std::vector<unsigned long long> ts((size_t)nb * 16);
            (void)hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_ws_ts), ts.size() * sizeof(unsigned long long));
            for (int b2 = 0; b2 < 4 && b2 + 256 < nb; ++b2) {
                const unsigned long long base = ts[(size_t)b2 * 16];
                fprintf(stderr, "blk %d   :", b2);
                for (int i = 0; i < 16; ++i) fprintf(stderr, " %lld", (long long)(ts[(size_t)b2 * 16 + i] - base));
                fprintf(stderr, "\nblk %d :", b2 + 256);
                for (int i = 0; i < 16; ++i) fprintf(stderr, " %lld", (long long)(ts[(size_t)(b2 + 256) * 16 + i] - base));
                fprintf(stderr, "\n");
            }
        }
        if (getenv("IODINE_WS_SE")) {                         // start / end of every block relative to the first start; pairs sharing a CU
            std::vector<unsigned long long> se((size_t)nb * 2);
            (void)hipMemcpyFromSymbol(se.data(), HIP_SYMBOL(g_ws_se), se.size() * sizeof(unsigned long long));
            unsigned long long t0 = ~0ull;
            for (int b2 = 0; b2 < nb; ++b2) t0 = std::min(t0, se[(size_t)b2 * 2]);
            std::vector<long long> st_(nb), en_(nb);
            for (int b2 = 0; b2 < nb; ++b2) { st_[b2] = (long long)(se[(size_t)b2 * 2] - t0); en_[b2] = (long long)(se[(size_t)b2 * 2 + 1] - t0); }
            std::vector<long long> es(en_), ss(st_);
            std::sort(es.begin(), es.end()); std::sort(ss.begin(), ss.end());
            fprintf(stderr, "[ws prof] block start: median %lld max %lld | block end: min %lld p25 %lld median %lld p75 %lld max %lld\n", ss[nb / 2], ss[nb - 1],
                    es[0], es[nb / 4], es[nb / 2], es[3 * nb / 4], es[nb - 1]);
            // pairs on the same CU (same xcc / se / sh / cu bits of HW_ID)
            int shown = 0;
            for (int a = 0; a < nb && shown < 6; ++a)
                for (int c2 = a + 1; c2 < nb; ++c2) {
                    const unsigned ha = hp[(size_t)a * 8 + 6], hc = hp[(size_t)c2 * 8 + 6];
                    if ((hp[(size_t)a * 8 + 7] & 0xf) == (hp[(size_t)c2 * 8 + 7] & 0xf) && ((ha >> 8) & 0xff) == ((hc >> 8) & 0xff)) {
                        fprintf(stderr, "  CU pair blocks %d / %d: start %lld / %lld end %lld / %lld (wave slots %u / %u)\n", a, c2, st_[a], st_[c2], en_[a], en_[c2],
                                ha & 0xf, hc & 0xf);
                        ++shown;
                        break;
                    }
                }
        }
        if (getenv("IODINE_WS_HWID")) {
            for (int b2 = 0; b2 < nb; ++b2) {
                const unsigned hw = hp[(size_t)b2 * 8 + 6], xc = hp[(size_t)b2 * 8 + 7];
                fprintf(stderr, "blk %d xcc %u se %u sh %u cu %u simd %u wave %u\n", b2, xc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf,
                        (hw >> 4) & 3, hw & 0xf);
            }
        }
    }
#endif
    return hipGetLastError();
}

hipError_t launch_conv3x3_ws_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias,
                                   const float* aux, float* out, const float* tmax_in, float* tmax_out, int N, int S, int c,
                                   int epi, int rev)
{
    // power-of-two image sizes only; the per-cell max of the INPUT (tmax_in, launch_cell_max or the producer's epilogue) is required
    if (S < 16 || (S & (S - 1)) != 0 || !tmax_in || (epi != EPI_L0ROWS && epi != EPI_L0ROWSX && !tmax_out)) return hipErrorInvalidValue;
#define WS_CASE(CC, EP) if (c == CC && epi == EP) return launch_ws_inst<CC, EP>(st, in, wpk, wmeta, bias, aux, out, tmax_in, tmax_out, N, S, rev);
    WS_CASE(64, EPI_BIAS_ELU) WS_CASE(64, EPI_MUL_ELUGRAD) WS_CASE(64, EPI_L0ROWS) WS_CASE(64, EPI_L0ROWSX)
    WS_CASE(32, EPI_BIAS_ELU) WS_CASE(32, EPI_MUL_ELUGRAD) WS_CASE(32, EPI_L0ROWS) WS_CASE(32, EPI_L0ROWSX)
#undef WS_CASE
    return hipErrorInvalidValue;
}

// ---- exact-fp32 form (option conv_precision 0): the same kernel with fp32 weights in the 144 registers and v_mfma_f32_16x16x4_f32 ----
// packed weights: [cout group][chunk of 32 cin][tap][half][lane][4 fp32]: lane l holds row = cout 16 cg + l % 16,
// cin 32 c + 16 half + 4 (l / 16) + j for the j-th MFMA over a 16-byte activation fragment (K index of that MFMA = l / 16)
__global__ void pack_conv_weights_ws32_kernel(const float* __restrict__ src, int C, int tflip, float* __restrict__ dst)
{
    const size_t total = (size_t)(C / 16) * (C / 32) * 9 * 2 * 64 * 4;
    const int nchunk = C / 32;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = idx & 3;
        size_t r = idx >> 2;
        const int lane = r & 63; r >>= 6;
        const int hl = r & 1; r >>= 1;
        const int tap = r % 9; r /= 9;
        const int c = r % nchunk;
        const int cg = (int)(r / nchunk);
        const int co = 16 * cg + (lane & 15), ci = 32 * c + 16 * hl + 4 * (lane >> 4) + j;
        // forward: W[co][ci][tap]; data gradient: the transposed conv, W[ci][co][8 - tap]
        dst[idx] = tflip ? src[((size_t)ci * C + co) * 9 + (8 - tap)] : src[((size_t)co * C + ci) * 9 + tap];
    }
}

hipError_t launch_pack_conv_weights_ws32(hipStream_t st, const float* src, int C, int tflip, void* dst)
{
    const size_t total = (size_t)(C / 16) * (C / 32) * 9 * 2 * 64 * 4;
    hipLaunchKernelGGL(pack_conv_weights_ws32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, C, tflip, (float*)dst);
    return hipGetLastError();
}

hipError_t launch_conv3x3_ws_f32(hipStream_t st, const float* in, const void* wpk, const float* bias, const float* aux, float* out,
                                 int N, int S, int c, int epi, int rev)
{
    if (S < 16 || (S & (S - 1)) != 0) return hipErrorInvalidValue;
#define WS_CASE(CC, EP) if (c == CC && epi == EP) return launch_ws_inst<CC, EP, true>(st, in, wpk, nullptr, bias, aux, out, nullptr, nullptr, N, S, rev);
    WS_CASE(64, EPI_BIAS_ELU) WS_CASE(64, EPI_MUL_ELUGRAD) WS_CASE(64, EPI_L0ROWS) WS_CASE(64, EPI_L0ROWSX)
    WS_CASE(32, EPI_BIAS_ELU) WS_CASE(32, EPI_MUL_ELUGRAD) WS_CASE(32, EPI_L0ROWS) WS_CASE(32, EPI_L0ROWSX)
#undef WS_CASE
    return hipErrorInvalidValue;
}
