// EXPERIMENT v2 (round 4; v1 = kernels_wino.hip, round 3; record: profiles/r04_winograd.md).  Changes against v1:
//   (1) the input halo of a unit (6 x 18 pixels x 64 channels, fp32) is loaded ONCE (four 16-byte loads per thread, one unit ahead) and
//       staged through LDS; the 4 x 4 patches are read from there (v1: seventeen 8-byte patch loads per thread = 2.4x the halo bytes
//       through the vector-memory path, measured 1.2 - 1.9 k cycles of load issue per unit);
//   (2) ONE V buffer instead of two (the halo buffer takes its place): the phases of a unit run in LOCKSTEP over all eight waves -
//       [output transform of unit s-1 ; input transform of unit s] barrier [48 MFMAs per wave ; halo of unit s+1 -> LDS] barrier -
//       each phase bound by ONE resource (VALU + LDS stores / matrix pipe + LDS reads) instead of two waves of a SIMD contending
//       for both in opposite orders.
// EXPERIMENT (round 3, NOT part of libiodine_hip.so - see profiles/r03_winograd.md for the measured go / no-go: correct, but 0.83-0.93 ms
// per cfg3 launch against 0.66-0.68 ms of the weight-stationary direct kernel).  Built only by tools/wino_variants.sh into
// iodine_amd/ab/libwino_<name>.so (api compiled with -DIODINE_WITH_WINO: op mode 11); checked by tools/experiments/wino_check.py.
//
// Winograd F(2x2, 3x3) form of the decoder's stride-1 3x3 conv 64 -> 64 (forward + bias + ELU and the data gradient with
// transposed / flipped weights x ELU'), split-fp16 arithmetic like kernels_convws.hip (fp32 operands as fp16 hi + lo, three
// MFMAs, fp32 accumulate).  Reference: nn.Conv2d + F.elu of MultiLayerConv, lib/modeling/iodine.py:583-592, and its autograd.
//
// Why: the weight-stationary direct kernel sits at the chip's power envelope (matrix-pipe busy x clock = 0.47
// of the 2.4 GHz peak whatever the instruction stream looks like), so only fewer matrix FLOPs per output move it.  F(2x2,3x3)
// computes a 2x2 output tile from a 4x4 input patch with 16 instead of 36 multiplies per (cin, cout): 12 instead of 27 split
// passes per output pixel-channel (2.25x fewer MFMAs):
//     U = G g G^T (per cout, cin; once per set_params)     V = B^T d B (per tile, cin)     M_xi = U_xi . V_xi  (xi = 16 positions,
//     a [64 cout x 64 cin] . [64 cin x tiles] GEMM each)    Y = A^T M A (per tile, cout), + bias, ELU  resp.  x ELU'(aux).
// Transforms run in fp32 on the VALU (+-1 / +-1/2 coefficients), the split into fp16 hi + lo happens AFTER them (V) resp. at pack
// time (U), so the three MFMA passes see the same kind of operands as in the direct kernels; measured error vs fp64 is that of an
// fp32 convolution (tools/experiments/wino_precision_sim.py: 3.6e-7 rel-L2, direct split 2.1e-7, ATen fp32 2.2e-7).
//
// Organisation (one persistent 8-wave block per CU):
//   * the transformed weights (16 xi x 64 x 64, hi + lo = 262 KB) live in registers: wave w owns xi = 2w, 2w+1 for ALL 64 cout
//     and 64 cin = 128 VGPRs, loaded once per block;
//   * a UNIT is 16 Winograd tiles = 4 x 16 output pixels (2 x 8 tiles, 6 x 18 input halo).  LDS holds V of a unit as
//     [xi][tile][hi 64 cin | lo 64 cin] at a 272-byte tile stride (conflict-free 16-byte fragment reads), 68 KB, double-buffered;
//     wave w reads only ITS xi regions (D[16 cout x 16 tiles] += U[16 x 32 cin] . V[32 cin x 16 tiles], v_mfma_f32_16x16x32_f16,
//     24 MFMAs per xi) and writes M back IN PLACE over the V region it has just consumed (same size: 64 cout x 4 B);
//   * every thread transforms one (tile, cin pair) of the NEXT unit (raw 4 x 4 patch prefetched one unit ahead with raw buffer
//     loads, hardware zero fill outside the image) and one (tile, output row, 4 cout) of the PREVIOUS unit, between the MFMAs of
//     the current one.  Wave w transforms tiles 2w, 2w+1 in both directions, so the slots its input transform overwrites are the
//     ones its own output transform has just read: ONE barrier per unit;
//   * power-of-two tile scales come from the producer's per-cell max side buffer (kernels_convws.hip), |V| <= 4 max |x|.
#include "../../iodine_amd/csrc/common.h"
#include <cstdio>
#include <vector>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4w __attribute__((ext_vector_type(4)));
// LDS is accessed as fp16 fragments, fp32 accumulator quads and packed dwords AT THE SAME ADDRESSES (M is written in place over V,
// V of the next unit over M): every LDS access goes through may_alias types, or type-based alias analysis lets hipcc reorder e.g.
// the input transform's dword stores above the output transform's float4 loads of the same bytes.
typedef f16x8 __attribute__((may_alias)) lds_f16x8;
typedef f32x4 __attribute__((may_alias)) lds_f32x4;
typedef unsigned __attribute__((may_alias)) lds_u32;

constexpr int WN_TS = 272;                 // bytes per (xi, tile): 128 hi | 128 lo | 16 pad (resp. 64 fp32 cout)
constexpr int WN_RS = 16 * WN_TS;          // bytes per xi region (16 tiles)
constexpr int WN_BUF = 16 * WN_RS;         // one unit: 69632 bytes
constexpr int WN_HPX = 6 * 18;             // halo pixels of a unit
constexpr int WN_RAW = WN_HPX * 256;       // raw fp32 halo: 27648 bytes
constexpr int WN_LDS = WN_BUF + WN_RAW + 128;   // + per-wave output maxima [2][8] floats (+ pad)

// U[xi = a*4+b][co][ci] = sum_ij G[a][i] g[co][ci][i][j] G[b][j]; tflip: the data-gradient conv (g'[co][ci][i][j] = g[ci][co][2-i][2-j])
__global__ void wino_weight_transform_kernel(const float* __restrict__ w, int C, int tflip, float* __restrict__ U)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * C) return;
    const int co = idx / C, ci = idx % C;
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            g[i][j] = tflip ? w[((size_t)ci * C + co) * 9 + (2 - i) * 3 + (2 - j)] : w[((size_t)co * C + ci) * 9 + i * 3 + j];
    float t[4][3];                                                            // G g
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
        U[((size_t)(a * 4 + 0) * C + co) * C + ci] = u0;
        U[((size_t)(a * 4 + 1) * C + co) * C + ci] = u1;
        U[((size_t)(a * 4 + 2) * C + co) * C + ci] = u2;
        U[((size_t)(a * 4 + 3) * C + co) * C + ci] = u3;
    }
}

__global__ __launch_bounds__(1024) void wino_weight_scale_kernel(const float* __restrict__ U, int n, float* __restrict__ meta)
{
    __shared__ float s_red[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(U[i]));
    m = wave_max_f32(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = 0.f;
        for (int k = 0; k < 16; ++k) mx = fmaxf(mx, s_red[k]);
        int e = 0;
        const bool ok = mx > 0.f && isfinite(mx);
        if (ok) frexpf(mx, &e);
        meta[0] = ok ? ldexpf(1.f, 13 - e) : 1.f;                   // max |U| * scale in [2^12, 2^13)
        meta[1] = 1.f / meta[0];
    }
}

// packed: [xi][chunk of 32 cin][cout group of 16][hi/lo][lane][8 fp16] = the A operand of v_mfma_f32_16x16x32_f16
// (lane l: row = cout 16 cg + l % 16, k = cin 32 c + 8 (l / 16) .. + 7)
__global__ void wino_weight_pack_kernel(const float* __restrict__ U, int C, const float* __restrict__ meta, _Float16* __restrict__ dst)
{
    const float scale = meta[0];
    const int nchunk = C / 32, ncg = C / 16;
    const size_t total = (size_t)16 * nchunk * ncg * 2 * 64 * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 7;
        size_t r = idx >> 3;
        const int lane = r & 63; r >>= 6;
        const int hl = r & 1; r >>= 1;
        const int cg = r % ncg; r /= ncg;
        const int c = r % nchunk;
        const int xi = (int)(r / nchunk);
        const int co = 16 * cg + (lane & 15), ci = 32 * c + 8 * (lane >> 4) + e;
        const float v = U[((size_t)xi * C + co) * C + ci] * scale;
        const _Float16 hi = (_Float16)v;
        dst[idx] = hl == 0 ? hi : (_Float16)(v - (float)hi);
    }
}

IOD_DEVINL float wino_fresh_scale(float mx)
{
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
    int se = 12 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return __uint_as_float((unsigned)(127 + se) << 23);
}

template <int I, int N, typename F>
IOD_DEVINL void wn_static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wn_static_for<I + 1, N>(f);
    }
}

#ifdef IODINE_TILE_PROF
__device__ unsigned g_wino_prof[TP_MAXBLK * 8];
#endif

template <int EPI>
__global__ __launch_bounds__(512)
void conv3x3_wino_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                               const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                               const float* __restrict__ tmax_in, float* __restrict__ tmax_out, int S, int lgS, int nunits, int rev)
{
    constexpr int C = 64;
    constexpr bool GRADF = EPI == EPI_MUL_ELUGRAD;
    static_assert(EPI == EPI_BIAS_ELU || EPI == EPI_MUL_ELUGRAD, "epilogue");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    unsigned char* s_raw = smem_w + WN_BUF;
    float* s_wmax = reinterpret_cast<float*>(smem_w + WN_BUF + WN_RAW);       // [2][8]

    using std::integral_constant;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    auto make_rsrc = [&](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)base;
        i32x4w r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
#define WN_BLOAD2(dst, voff, rsrc, imm) \
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:%3" : "=v"(dst) : "v"(voff), "s"(rsrc), "n"(imm))
#define WN_SGPR_SETTLE(rsrc) asm volatile("s_nop 4" :: "s"(rsrc))

    // ---- this wave's transformed-weight slice -> registers (once per block): xi = 2 wv + xl, chunk c, cout group cg ----
    f16x8 wh[2][2][4], wl[2][2][4];
    {
        const uint4* wp = wpk + (size_t)(2 * wv) * 2 * 4 * 2 * 64 + lane;
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const uint4 h = wp[(((xl * 2 + c) * 4 + cg) * 2 + 0) * 64], l = wp[(((xl * 2 + c) * 4 + cg) * 2 + 1) * 64];
                    __builtin_memcpy(&wh[xl][c][cg], &h, 16);
                    __builtin_memcpy(&wl[xl][c][cg], &l, 16);
                }
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) asm volatile("" : "+v"(wh[xl][c][cg]), "+v"(wl[xl][c][cg]));     // opaque: stay in registers
    }
    const float inv_w = wmeta[1];

    // ---- thread roles ----
    // MFMA: tile column n, k block / accumulator row block.  Input transform: (tile, cin pair).  Output transform: (tile, output
    // row, cout quad).  Wave w transforms tiles 2w, 2w+1 in BOTH directions: the output transform of unit s-1 reads exactly the
    // (xi, tile) slots its input transform of unit s+1 overwrites, so that hazard is wave-local and ONE barrier per unit is enough.
    const int mn = lane & 15, mkb = lane >> 4;
    const int in_n = tid >> 5, in_p = tid & 31;
    const int ity = in_n >> 3, itx = in_n & 7;
    const int oq = lane & 15, orow = (lane >> 4) & 1, on = 2 * wv + (lane >> 5);
    const int oty = on >> 3, otx = on & 7;
    const unsigned frag_off = (unsigned)((2 * wv) * WN_RS + mn * WN_TS + mkb * 16);       // + xl * RS + c * 64 (+128 lo) / + cg * 64
    const unsigned vst_off = (unsigned)(in_n * WN_TS + in_p * 4);                          // + xi * RS (+128 lo)
    const unsigned mrd_off = (unsigned)(on * WN_TS + oq * 16 + (orow * 4) * WN_RS);        // + xi * RS (first xi row used = orow)
    const float osg = orow ? -1.f : 1.f;

    // ---- persistent, XCD-aware schedule over the units (like kernels_convws.hip) ----
    const int units_x = S >> 4, units_y = S >> 2, lg_ux = lgS - 4, lg_upi = 2 * lgS - 6;
    const int cells_x = S >> 4, cells_y = S >> 3, lg_cpi = 2 * lgS - 7;
    const int nblk = gridDim.x;
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = (nblk + 7) >> 3;
    const int per_xcd = (nunits + 7) >> 3;
    const int u_begin = xcd * per_xcd, u_end = min(nunits, u_begin + per_xcd);
    const int u0 = u_begin + bix;
    const int nu = u0 < u_end ? (u_end - u0 + bpx - 1) / bpx : 0;
    if (nu == 0) return;
    auto unit_coords = [&](int s, int& n, int& uy, int& ux) {
        const int u = u0 + s * bpx;
        const int uu = rev ? nunits - 1 - u : u;
        n = uu >> lg_upi;
        uy = (uu >> lg_ux) & (units_y - 1); ux = uu & (units_x - 1);
    };

    // ---- halo of a unit: 108 pixels x 16 float4 = 1728 float4 over 512 threads (k = 0..3), loaded one unit ahead into registers
    // (plain loads, hardware zero fill through an out-of-range buffer offset), stored to LDS in the MFMA phase ----
    constexpr int NRAW = 5, NOST = 2;                       // vector-memory ops per issue_raw (4 halo + 1 side buffer); output stores
    // (deliberately NOT initialised: an initial value would be a second reaching definition of these loop-carried asm destinations)
    f32x4 hal[4];
    float tmv;
    f32x4 ax0, ax1;
    (void)ax0; (void)ax1;
    unsigned hvoff[4], hflg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 512 * k;
        const int px = idx >> 4, seg = idx & 15, hy = px / 18, hx = px % 18;
        hvoff[k] = (unsigned)((((hy << lgS) + hx) * C + seg * 4) * 4);
        hflg[k] = (idx < WN_HPX * 16 ? 16u : 0u) | (hy == 0 ? 1u : 0u) | (hy == 5 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == 17 ? 8u : 0u);
    }
    const int cellq = min(lane >> 2, 8);
#define WN_BLOAD4(dst, voff, rsrc) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc))
    auto issue_raw = [&](int s) {
        const bool live = s < nu;
        int n, uy, ux;
        unit_coords(live ? s : 0, n, uy, ux);
        // descriptor base = pixel (4 uy - 1, 16 ux - 1) of the slot-image (outside for uy = 0 / ux = 0: those lanes are masked)
        const float* org = in + (size_t)n * S * S * C + ((long long)(uy * 4 - 1) * S + ux * 16 - 1) * C;
        const i32x4w rsrc = make_rsrc(org, 0x7fffffffu);
        const unsigned edge = (uy == 0 ? 1u : 0u) | (uy == units_y - 1 ? 2u : 0u) | (ux == 0 ? 4u : 0u) | (ux == units_x - 1 ? 8u : 0u);
        unsigned vo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) vo[k] = (live && (hflg[k] & 16u) && !(hflg[k] & edge)) ? hvoff[k] : 0x80000000u;
        const int cy = min(max((uy >> 1) + cellq / 3 - 1, 0), cells_y - 1), cx = min(max(ux + cellq % 3 - 1, 0), cells_x - 1);
        const float* p = tmax_in + ((((size_t)n << lg_cpi) + (size_t)(cy * cells_x) + cx) << 2) + (lane & 3);
        WN_SGPR_SETTLE(rsrc);
#pragma unroll
        for (int k = 0; k < 4; ++k) WN_BLOAD4(hal[k], vo[k], rsrc);
        asm volatile("global_load_dword %0, %1, off" : "=v"(tmv) : "v"(p));
    };
    auto tie_raw = [&]() {                                  // the loaded registers become visible to the compiler here
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(hal[k]));
        asm volatile("" : "+v"(tmv));
    };
    auto store_halo = [&]() {                               // registers -> raw LDS halo [px][64 ch] fp32
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (hflg[k] & 16u) *reinterpret_cast<lds_f32x4*>(s_raw + (size_t)(tid + 512 * k) * 16) = hal[k];
    };
// n = vector-memory operations issued after the raw loads waited for
#define WN_WAIT_RAW(n) do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n)); tie_raw(); } while (0)
    // output pixel of this thread in unit s: byte offset inside the slot-image, and the image's buffer descriptors
    auto out_voff = [&](int uy, int ux) {
        const int y = uy * 4 + 2 * oty + orow, x0 = ux * 16 + 2 * otx;
        return (unsigned)(((((y << lgS) + x0) * C) << 2) + (oq << 4));
    };
    auto issue_aux = [&](int s) {
        if constexpr (GRADF) {
            int n, uy, ux;
            unit_coords(s, n, uy, ux);
            const i32x4w rsrc = make_rsrc(aux + (size_t)n * S * S * C, (unsigned)(S * S * C * 4));
            const unsigned vo = out_voff(uy, ux);
            WN_SGPR_SETTLE(rsrc);
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ax0) : "v"(vo), "s"(rsrc));
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:256" : "=v"(ax1) : "v"(vo), "s"(rsrc));
        }
    };
    // V = B^T d B of this thread's (tile, cin pair), split into fp16 hi / lo at `scale`, into LDS buffer `buf`
    typedef f32x2 __attribute__((may_alias)) lds_f32x2;
    const unsigned prd_off = (unsigned)(((2 * ity) * 18 + 2 * itx) * 256 + in_p * 8);
    auto transform_store = [&](unsigned char* buf, float scale) {
        f32x2 raw[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[i][j] = *reinterpret_cast<const lds_f32x2*>(s_raw + prd_off + (i * 18 + j) * 256);
        f32x2 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = raw[0][j] - raw[2][j];
            t[1][j] = raw[1][j] + raw[2][j];
            t[2][j] = raw[2][j] - raw[1][j];
            t[3][j] = raw[1][j] - raw[3][j];
        }
        unsigned char* dst = buf + vst_off;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x2 v[4];
            v[0] = t[a][0] - t[a][2];
            v[1] = t[a][1] + t[a][2];
            v[2] = t[a][2] - t[a][1];
            v[3] = t[a][1] - t[a][3];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x2 x = v[b] * scale;
                f32x2 h;
                h.x = __uint_as_float(__float_as_uint(x.x) & 0xffffe000u); h.y = __uint_as_float(__float_as_uint(x.y) & 0xffffe000u);
                const f32x2 l = x - h;
                typedef __fp16 h2 __attribute__((ext_vector_type(2)));
                const h2 hh = __builtin_amdgcn_cvt_pkrtz(h.x, h.y), ll = __builtin_amdgcn_cvt_pkrtz(l.x, l.y);
                unsigned uh, ul;
                __builtin_memcpy(&uh, &hh, 4); __builtin_memcpy(&ul, &ll, 4);
                *reinterpret_cast<lds_u32*>(dst + (a * 4 + b) * WN_RS) = uh;
                *reinterpret_cast<lds_u32*>(dst + (a * 4 + b) * WN_RS + 128) = ul;
            }
        }
    };
    // M_xi = U_xi . V_xi for this wave's xi = 2 wv + xl from LDS buffer `buf`, written back in place
    auto mfma_xi = [&](auto xlc, unsigned char* buf) {
        constexpr int xl = decltype(xlc)::value;
        unsigned char* src = buf + frag_off + xl * WN_RS;
        f16x8 vh[2], vlo[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            vh[c] = *reinterpret_cast<const lds_f16x8*>(src + c * 64);
            vlo[c] = *reinterpret_cast<const lds_f16x8*>(src + c * 64 + 128);
        }
        f32x4 acc[4];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[xl][c][cg], vlo[c], acc[cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[xl][c][cg], vh[c], acc[cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[xl][c][cg], vh[c], acc[cg], 0, 0, 0);
        }
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) *reinterpret_cast<lds_f32x4*>(src + cg * 64) = acc[cg];
    };
    f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_BIAS_ELU) {
        const float4 tb = *reinterpret_cast<const float4*>(bias + oq * 4);
        b4 = f32x4{tb.x, tb.y, tb.z, tb.w};
    }
    // Y = A^T M A of this thread's (tile, output row, 4 cout) of unit s from LDS buffer `buf`; epilogue; stores
    auto output_transform = [&](int s, const unsigned char* buf, float inv, auto aux_younger) {
        int n, uy, ux;
        unit_coords(s, n, uy, ux);
        const i32x4w rsrc = make_rsrc(out + (size_t)n * S * S * C, (unsigned)(S * S * C * 4));
        const unsigned vo = out_voff(uy, ux);
        const unsigned char* src = buf + mrd_off;
        f32x4 o0 = f32x4{0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 m0 = *reinterpret_cast<const lds_f32x4*>(src + (0 * 4 + b) * WN_RS);
            const f32x4 m1 = *reinterpret_cast<const lds_f32x4*>(src + (1 * 4 + b) * WN_RS);
            const f32x4 m2 = *reinterpret_cast<const lds_f32x4*>(src + (2 * 4 + b) * WN_RS);
            const f32x4 R = m0 + osg * (m1 + m2);                             // row 0: M0 + M1 + M2, row 1: M1 - M2 - M3
            if (b == 0) o0 = R;
            else if (b == 1) { o0 += R; o1 = R; }
            else if (b == 2) { o0 += R; o1 -= R; }
            else o1 -= R;
        }
        o0 *= inv; o1 *= inv;
        if constexpr (EPI == EPI_BIAS_ELU) {
            o0 += b4; o1 += b4;
            o0 = f32x4{elu1_fast(o0.x), elu1_fast(o0.y), elu1_fast(o0.z), elu1_fast(o0.w)};
            o1 = f32x4{elu1_fast(o1.x), elu1_fast(o1.y), elu1_fast(o1.z), elu1_fast(o1.w)};
        } else {
#ifdef WN_DBG_AUXWAIT0
            asm volatile("s_waitcnt vmcnt(0)");
#else
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(decltype(aux_younger)::value));
#endif
            asm volatile("" : "+v"(ax0), "+v"(ax1));
            o0.x *= elu1_grad_from_out(ax0.x); o0.y *= elu1_grad_from_out(ax0.y); o0.z *= elu1_grad_from_out(ax0.z); o0.w *= elu1_grad_from_out(ax0.w);
            o1.x *= elu1_grad_from_out(ax1.x); o1.y *= elu1_grad_from_out(ax1.y); o1.z *= elu1_grad_from_out(ax1.z); o1.w *= elu1_grad_from_out(ax1.w);
        }
        WN_SGPR_SETTLE(rsrc);
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" :: "v"(o0), "v"(vo), "s"(rsrc));
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:256\n\ts_nop 1" :: "v"(o1), "v"(vo), "s"(rsrc));
        float vmax = fmaxf(fmaxf(fmaxf(fabsf(o0.x), fabsf(o0.y)), fmaxf(fabsf(o0.z), fabsf(o0.w))),
                           fmaxf(fmaxf(fabsf(o1.x), fabsf(o1.y)), fmaxf(fabsf(o1.z), fabsf(o1.w))));
        vmax = wave_max_f32(vmax);
        if (lane == 0) s_wmax[(s & 1) * 8 + wv] = vmax;
    };
    // per-cell max of the OUTPUT (side buffer for its consumer): this unit is the upper / lower half of an 8 x 16 cell and
    // fills two of the cell's four floats
    auto store_tmax = [&](int s) {
        if (tid < 2) {
            int n, uy, ux;
            unit_coords(s, n, uy, ux);
            const float* m = s_wmax + (s & 1) * 8;
            const float v = fmaxf(fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])), fmaxf(fmaxf(m[4], m[5]), fmaxf(m[6], m[7])));
            float* p = tmax_out + ((((size_t)n << lg_cpi) + (size_t)((uy >> 1) * cells_x) + ux) << 2) + 2 * (uy & 1) + tid;
            asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v));
        }
    };
    auto unit_scale = [&]() { return wino_fresh_scale(4.f * wave_max_f32(lane < 36 ? tmv : 0.f)); };   // |V| <= 4 max |x|

    TP_DECL;
    // The loop starts two iterations early (s = -2: request the halo of unit 0; s = -1: halo 0 -> LDS, request unit 1), so that the
    // inline-asm load destinations have ONE definition, inside the loop: with a second one in a prologue hipcc may copy the
    // loop-carried registers before the data has arrived.
    float next_scale = 1.f, cur_scale = 1.f, prev_scale = 1.f;
    unsigned char* V = smem_w;
    for (int s = -2; s < nu; ++s) {
        // ---- phase CA: output transform of unit s-1 (reads M from V), input transform of unit s (halo -> V).  A wave transforms the
        // same two tiles in both directions, so the slots it overwrites are the ones it has just read (v1's invariant) ----
        TP_STAMP(0);
        if (s >= 0) {
            if (s > 1) store_tmax(s - 2);
            if (s > 0) output_transform(s - 1, V, inv_w / prev_scale, integral_constant<int, NRAW>{});    // aux (s-1) is followed by halo (s+1)
            TP_STAMP(1);
            prev_scale = cur_scale = next_scale;
#ifdef WN_DBG_BARRIER
            __syncthreads();
#endif
            transform_store(V, cur_scale);
        }
        issue_aux(max(s, 0));
        TP_STAMP(2);
        __syncthreads();                                     // V_s complete; the halo buffer is free
        TP_STAMP(3);
        // ---- phase B: 48 MFMAs per wave (M over V in place); halo of unit s+1: registers -> LDS, halo of unit s+2 requested ----
        if (s >= 0) mfma_xi(integral_constant<int, 0>{}, V);
        if (s + 1 >= 0 && s + 1 < nu) {
            if (s > 0) { if constexpr (GRADF) WN_WAIT_RAW(NOST + 2); else WN_WAIT_RAW(NOST); }     // halo (s+1) is followed by 2 stores [, 2 aux loads]
            else WN_WAIT_RAW(0);
            next_scale = unit_scale();
            store_halo();
        }
        if (s >= 0) mfma_xi(integral_constant<int, 1>{}, V);
        issue_raw(s + 2);
        TP_STAMP(4);
        __syncthreads();                                     // M_s complete, halo of unit s+1 in LDS
        TP_STAMP(5);
    }
    // the loads issued for the (non-existent) units nu, nu + 1 are still in flight: their destination registers must stay reserved
    WN_WAIT_RAW(0);
    if (nu > 1) store_tmax(nu - 2);
    output_transform(nu - 1, V, inv_w / prev_scale, integral_constant<int, 0>{});
    __syncthreads();
    store_tmax(nu - 1);
#ifdef IODINE_TILE_PROF
    if (tid == 0 && blockIdx.x < TP_MAXBLK) for (int i_ = 0; i_ < 8; ++i_) g_wino_prof[blockIdx.x * 8 + i_] = tp_acc[i_];
#endif
#undef WN_BLOAD2
#undef WN_SGPR_SETTLE
#undef WN_WAIT_RAW
}

template <int EPI>
hipError_t launch_wino_inst(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias, const float* aux,
                            float* out, const float* tmax_in, float* tmax_out, int N, int S, int rev)
{
    static std::atomic<unsigned> attr_devs{0};
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_wino_f16x3_kernel<EPI>, WN_LDS, attr_devs); e != hipSuccess) return e;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    int lgS = 0;
    while ((1 << lgS) < S) ++lgS;
    const int nunits = N * (S / 16) * (S / 4);
    const int per_xcd = (nunits + 7) / 8;
    const int bpx = std::min(per_xcd, std::max(1, n_cu / 8));                  // one persistent 8-wave block per CU
    hipLaunchKernelGGL((conv3x3_wino_f16x3_kernel<EPI>), dim3(8 * bpx), dim3(512), WN_LDS, st, in, reinterpret_cast<const uint4*>(wpk),
                       wmeta, bias, aux, out, tmax_in, tmax_out, S, lgS, nunits, rev);
#ifdef IODINE_TILE_PROF
    {
        const int nb = std::min(8 * bpx, TP_MAXBLK);
        std::vector<unsigned> hp((size_t)nb * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_wino_prof), hp.size() * sizeof(unsigned));
        static const char* names[8] = {"loop", "output transform", "input transform", "barrier 1", "mfma + halo store + load issue", "barrier 2", "-", "-"};
        double sum[8] = {0};
        for (int b2 = 0; b2 < nb; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[(size_t)b2 * 8 + i];
        fprintf(stderr, "[wino prof] memtime ticks per UNIT (wave 0, %d units per block):", (nunits + 8 * bpx - 1) / (8 * bpx));
        for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.0f |", names[i], sum[i] / (double)nunits);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

}  // namespace

size_t conv_wino_wpk_bytes(int C) { return (size_t)16 * (C / 32) * (C / 16) * 2 * 64 * 16; }
size_t conv_wino_scratch_floats(int C) { return (size_t)16 * C * C; }

// transformed + packed weights of one layer / direction; `scratch` holds conv_wino_scratch_floats(C) floats
hipError_t launch_pack_conv_weights_wino(hipStream_t st, const float* src, int C, int tflip, float* meta, void* dst, float* scratch)
{
    if (C != 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(wino_weight_transform_kernel, dim3((C * C + 255) / 256), dim3(256), 0, st, src, C, tflip, scratch);
    hipLaunchKernelGGL(wino_weight_scale_kernel, dim3(1), dim3(1024), 0, st, scratch, 16 * C * C, meta);
    const size_t total = conv_wino_wpk_bytes(C) / 2;
    hipLaunchKernelGGL(wino_weight_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, scratch, C, meta, (_Float16*)dst);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias,
                                     const float* aux, float* out, const float* tmax_in, float* tmax_out, int N, int S, int c,
                                     int epi, int rev)
{
    if (c != 64 || S < 16 || (S & (S - 1)) != 0 || !tmax_in || !tmax_out) return hipErrorInvalidValue;
    if (epi == EPI_BIAS_ELU) return launch_wino_inst<EPI_BIAS_ELU>(st, in, wpk, wmeta, bias, aux, out, tmax_in, tmax_out, N, S, rev);
    if (epi == EPI_MUL_ELUGRAD) return launch_wino_inst<EPI_MUL_ELUGRAD>(st, in, wpk, wmeta, bias, aux, out, tmax_in, tmax_out, N, S, rev);
    return hipErrorInvalidValue;
}
