// Weight-stationary form of the refinement network's stride-2 3x3 convs C -> C (layers 1 .. of RefinementNetwork.mlc,
// lib/modeling/iodine.py:459,480: conv k3 s2 p1 + ELU), split-fp16 arithmetic (fp32 operands as fp16 hi + lo, three MFMAs, fp32
// accumulate - see kernels_conv.hip).  gfx950 only.  Round 4.
//
// Why another kernel: conv3x3_s2_f16x3_kernel (kernels_refine.hip) walks 16 stages (4 parity sub-images x 4 channel chunks) per 16 x 16
// tile, each with two barriers and its own slice of the packed weights staged into LDS - 147 KB of weights per tile through L2 and LDS,
// as many bytes as the activations of layers 2 and 3 - in 3.5 non-persistent blocks per CU: 0.11 / 0.03 / 0.03 ms per iteration at
// cfg3 for 0.29 / 0.07 / 0.02 GB (2.7 TB/s and less).  Here, as in kernels_convws.hip:
//   * a wave OWNS 16 output channels and keeps their whole [C x 9 taps x 16] weight slice, hi and lo, in 144 VGPRs for the life of a
//     persistent block (two blocks per CU);
//   * a tile is 2 x 16 output pixels; LDS holds only the input halo of one 32-channel chunk (5 fine rows x 33 fine columns, 160-byte
//     pixel stride as in the stride-1 kernel) with the columns DE-INTERLEAVED by parity - output column X needs fine column
//     2 X + kx - 1, i.e. consecutive positions of the even plane (kx = 0: X, kx = 2: X + 1) or of the odd plane (kx = 1) - so the 16 lanes
//     of a fragment read hit 16 distinct 16-byte slots exactly like a stride-1 read;
//   * D[16 cout x 16 px] += W[16 x 32 cin] . X[32 x 16 px] with v_mfma_f32_16x16x32_f16; a fragment of fine row r feeds the one or two
//     (output row, ky) pairs with 2 y + ky = r: 54 MFMAs and 30 ds_read_b128 per wave, tile and chunk;
//   * the chunk's loads are issued TWO stages ahead (two register sets, inline-asm loads with counted vmcnt waits); the power-of-two scale of a
//     chunk is the block-wide max of the staged values (one extra barrier per chunk: the tensors these layers read have no per-cell side
//     buffer), accumulators are rescaled exactly when it changes between the two chunks of a tile.
#include "common.h"
#include <utility>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int RW_HR = 5, RW_HC = 33, RW_NPX = RW_HR * RW_HC;   // halo of a 2 x 16 output tile, fine pixels
constexpr int RW_PXB = 160;                                    // bytes per staged pixel: 64 hi | 64 lo | 32 pad
constexpr int RW_BUFB = (RW_NPX + 1) * RW_PXB;                 // + dump slot for idle lanes
constexpr int RW_NIN = (RW_NPX * 8 + 255) / 256;               // float4 loads per thread and chunk (6)

typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
IOD_DEVINL void rw_static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rw_static_for<I + 1, N>(f);
    }
}

IOD_DEVINL float rw_fresh_scale(float mx)
{
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
    int se = 12 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return __uint_as_float((unsigned)(127 + se) << 23);
}

// F32 (conv_precision 0, round 5): the staged pixel is its 32 fp32 channels (128 of the same 160 bytes), the 144 weight registers hold fp32
// (launch_pack_conv_weights_ws32: element j of a lane's 16 bytes = the A operand of the j-th v_mfma_f32_16x16x4_f32 over the fragment's 16
// bytes), 8 MFMAs per fragment and (row, ky) pair instead of 3; no scales: the max reduction and the accumulator rescale drop out, the barrier
// that frees the LDS buffer stays.
template <int C, bool F32 = false>
__global__ __launch_bounds__(256, 2)
void conv3x3_s2ws_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                               const float* __restrict__ bias, float* __restrict__ out, int Sc, int tiles_x, int tiles_y, int ntiles)
{
    static_assert(C == 64, "one wave per group of 16 output channels");
    constexpr int NCHUNK = C / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rw[];
    float* s_max = reinterpret_cast<float*>(smem_rw + 2 * RW_BUFB);          // [2 parity][4 waves]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wv;
    const int lpx = lane & 15, lkb = lane >> 4;
    const int Sf = 2 * Sc;

    // ---- this wave's weight slice -> registers (once per block) ----
    using wreg_t = std::conditional_t<F32, f32x4, f16x8>;
    wreg_t wh[NCHUNK][9], wl[NCHUNK][9];
    {
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
            for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(wh[c][t]), "+v"(wl[c][t]));
    }
    float inv_w = 1.f;
    if constexpr (!F32) inv_w = wmeta[1];
    float* s_bias = s_max + 8;                                               // [C] (kept out of the register file: 254 VGPRs are spoken for)
    if (tid < C) s_bias[tid] = bias[tid];

    // ---- staging: float4 k of this thread = halo pixel (tid >> 3) + 32 k (row r, halo column h), channel quad tid & 7 of the chunk: byte
    // offset relative to the halo's origin pixel (fine row 4 ty - 1, fine column 32 tx - 1), LDS slot with the columns de-interleaved
    // (even h -> position h / 2, odd h -> 17 + h / 2).  Recomputed per stage (a dozen VALU per float4) instead of kept in 18 VGPRs: the
    // registers pay for a SECOND set of staged values, i.e. loads that are two stages ahead instead of one ----
    const int px0 = tid >> 3, q8 = tid & 7;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_rw;
    const unsigned fr_base = lds_base + (unsigned)(lpx * RW_PXB + lkb * 16);

    const int nblk = gridDim.x;
    const int nt_blk = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + nblk - 1) / nblk : 0;
    const int nstage = nt_blk * NCHUNK;
    if (nstage == 0) return;

    // All vector-memory traffic of the loop is inline asm with COUNTED vmcnt waits: with compiler-tracked loads hipcc hoisted the next
    // stage's max reduction into the MFMA phase behind vmcnt(5) .. vmcnt(0) - i.e. waited for the loads it had just issued (back-edge
    // conservatism), which exposed a full HBM round trip per stage.  Loads and stores are issued for EVERY stage / tile (all lanes out of
    // range past the end), so the counts are uniform: in front of a stage's first use there are 6 younger loads + 2 younger stores in
    // flight (6 loads in the first iteration).
    typedef int i32x4r __attribute__((ext_vector_type(4)));
    auto make_rsrc = [&](const void* base) {
        const unsigned long long p = (unsigned long long)base;
        i32x4r r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        r.z = 0x7fffffff;
        r.w = 0x00020000;
        return r;
    };
    f32x4 rin[NCHUNK][RW_NIN];                                // [chunk]: stage s lives in set s % NCHUNK, requested two stages ahead
    auto issue = [&](int s, f32x4 (&rr)[RW_NIN]) {            // loads of stage s = (tile, chunk): block-uniform scalar part + per-thread part
        const bool live = s < nstage;
        const int t = (int)blockIdx.x + ((live ? s : 0) / NCHUNK) * nblk, c = s % NCHUNK;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const float* org = in + (size_t)n * Sf * Sf * C + ((long long)(4 * ty - 1) * Sf + 32 * tx - 1) * C + c * 32;
        const i32x4r rs = make_rsrc(org);
        const int rmax = live ? Sf - (4 * ty - 1) : 0, cmax = Sf - (32 * tx - 1);           // rows / columns of the halo inside the image
        const bool top = ty == 0, left = tx == 0;
        unsigned vo[RW_NIN];
#pragma unroll
        for (int k = 0; k < RW_NIN; ++k) {
            const int px = px0 + 32 * k;
            const int r = (px * 1986) >> 16, hcol = px - r * RW_HC;                  // px / 33 for px < 2048
            const bool inval = px >= RW_NPX || (top && r == 0) || (left && hcol == 0) || r >= rmax || hcol >= cmax;
            vo[k] = inval ? 0x80000000u : (unsigned)(((r * Sf + hcol) * C + q8 * 4) * 4);
        }
        asm volatile("s_nop 4" :: "s"(rs) : "memory");
#pragma unroll
        for (int k = 0; k < RW_NIN; ++k)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rr[k]) : "v"(vo[k]), "s"(rs) : "memory");
    };
    auto wait_stage = [&](bool first, f32x4 (&rr)[RW_NIN]) {  // the staged values become visible to the compiler here
        if (first) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RW_NIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RW_NIN + 2) : "memory");
#pragma unroll
        for (int k = 0; k < RW_NIN; ++k) asm volatile("" : "+v"(rr[k]));
    };
    auto convert = [&](int buf, float scale, const f32x4 (&rr)[RW_NIN]) {
        unsigned char* sb = smem_rw + buf * RW_BUFB;
#pragma unroll
        for (int k = 0; k < RW_NIN; ++k) {
            const int px = px0 + 32 * k;
            const int r = (px * 1986) >> 16, hcol = px - r * RW_HC;
            const int pos = r * RW_HC + ((hcol & 1) ? 17 + (hcol >> 1) : (hcol >> 1));
            if constexpr (F32) {                             // exact fp32: the float4 goes to LDS as it is (channel quad q8 at byte 16 q8)
                const int lo32 = px < RW_NPX ? pos * RW_PXB + q8 * 16 : RW_NPX * RW_PXB;
                *reinterpret_cast<f32x4*>(sb + lo32) = rr[k];
                continue;
            }
            const int lo_ = px < RW_NPX ? pos * RW_PXB + q8 * 8 : RW_NPX * RW_PXB;
            f32x4 v = rr[k] * scale;
            unsigned l0, l1;
            const unsigned h0 = pack_hi_lo(v.x, v.y, l0), h1 = pack_hi_lo(v.z, v.w, l1);
            *reinterpret_cast<uint2*>(sb + lo_) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(sb + lo_ + 64) = make_uint2(l0, l1);
        }
    };

#define RW_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    struct Frag { wreg_t h, l; };
    f32x4 acc[2];
    // all taps of chunk c from LDS buffer `buf`: fine halo rows r = 0..4, column taps kx = 0..2
    auto compute = [&](auto cc, int buf) {
        constexpr int c = decltype(cc)::value;
        const unsigned base = fr_base + (unsigned)(buf * RW_BUFB);
        Frag f[2];
        auto LOADF = [](auto sc, Frag& fr, unsigned b) {
            constexpr int s = decltype(sc)::value;
            constexpr int r = s / 3, kx = s % 3;
            constexpr int off = (r * RW_HC + (kx == 1 ? 17 : (kx == 2 ? 1 : 0))) * RW_PXB;
            RW_DSR128(fr.h, b, off);
            RW_DSR128(fr.l, b, off + 64);
        };
        auto MMA = [&](auto sc, const Frag& fr) {
            constexpr int s = decltype(sc)::value;
            constexpr int r = s / 3, kx = s % 3;
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int ky = r - 2 * y;
                if (ky >= 0 && ky <= 2) {
                    if constexpr (F32) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[y] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[c][ky * 3 + kx][j], fr.h[j], acc[y], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[y] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[c][ky * 3 + kx][j], fr.l[j], acc[y], 0, 0, 0);
                    } else {
                    acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c][ky * 3 + kx], fr.l, acc[y], 0, 0, 0);
                    acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c][ky * 3 + kx], fr.h, acc[y], 0, 0, 0);
                    acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c][ky * 3 + kx], fr.h, acc[y], 0, 0, 0);
                    }
                }
            }
        };
        LOADF(std::integral_constant<int, 0>{}, f[0], base);
        rw_static_for<0, 15>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s + 1 < 15) {
                LOADF(std::integral_constant<int, (s + 1 < 15 ? s + 1 : 0)>{}, f[(s + 1) & 1], base);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("" : "+v"(f[s & 1].h), "+v"(f[s & 1].l));
            MMA(sc, f[s & 1]);
        });
    };
#undef RW_DSR128

    rw_static_for<0, NCHUNK>([&](auto cc) { issue(decltype(cc)::value, rin[decltype(cc)::value]); });
    float tile_scale_cur = 1.f;
    for (int s0 = 0; s0 < nstage; s0 += NCHUNK) {
        rw_static_for<0, NCHUNK>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const int s = s0 + c, par = c & 1;
            // ---- stage s: max of the staged values -> block-wide scale -> split into LDS buffer c ----
            wait_stage(s0 == 0, rin[c]);
            float scale = 1.f;
            if constexpr (F32) {
                (void)par;
                __syncthreads();                              // every wave is done with the MFMAs of stage s - 2 = this buffer's last readers
            } else {
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < RW_NIN; ++k)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(rin[c][k].x), fabsf(rin[c][k].y)), fmaxf(fabsf(rin[c][k].z), fabsf(rin[c][k].w))));
            m = wave_max_f32(m);
            if (lane == 0) s_max[par * 4 + wv] = m;
            __syncthreads();                                  // (also: every wave is done with the MFMAs of stage s - 2 = this buffer's last readers)
            scale = rw_fresh_scale(fmaxf(fmaxf(s_max[par * 4], s_max[par * 4 + 1]), fmaxf(s_max[par * 4 + 2], s_max[par * 4 + 3])));
            }
            convert(c, scale, rin[c]);
            issue(s + NCHUNK, rin[c]);                            // two stages ahead, into the set just converted (masked past the end)
            if (c == 0) {
                acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
            } else if (scale != tile_scale_cur) {             // block-uniform; exact (powers of two)
                const float r = scale / tile_scale_cur;
                acc[0] *= r; acc[1] *= r;
            }
            tile_scale_cur = scale;
            __syncthreads();                                  // the chunk is complete in LDS
            compute(cc, c);
        });
        {
            // ---- epilogue: bias + ELU, 4 channels of one pixel per lane (16 lanes x 64 bytes per row and cout group) ----
            const int t = (int)blockIdx.x + (s0 / NCHUNK) * nblk;
            const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
            const float inv = inv_w / tile_scale_cur;
            const int X = 16 * tx + lpx;
            const i32x4r ro = make_rsrc(out + (size_t)n * Sc * Sc * C);
            const float4 bq = *reinterpret_cast<const float4*>(s_bias + 16 * cg + 4 * lkb);
            asm volatile("s_nop 4" :: "s"(ro) : "memory");
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int Y = 2 * ty + y;
                const f32x4 v = acc[y] * inv;
                const f32x4 o = f32x4{elu1_fast(v.x + bq.x), elu1_fast(v.y + bq.y), elu1_fast(v.z + bq.z), elu1_fast(v.w + bq.w)};
                const unsigned vo = (Y < Sc && X < Sc) ? (unsigned)(((Y * Sc + X) * C + 16 * cg + 4 * lkb) * 4) : 0x80000000u;
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" :: "v"(o), "v"(vo), "s"(ro) : "memory");
            }
        }
    }
    // the loads issued for the (non-existent) stages past the end are still in flight: their destination registers stay reserved
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int k = 0; k < RW_NIN; ++k) asm volatile("" : "+v"(rin[c][k]));
}

}  // namespace

bool conv3x3_s2ws_ok(int S, int c) { return c == 64 && S >= 4 && S % 2 == 0; }

// Forward stride-2 conv C -> C + bias + ELU, weight-stationary.  S = fine (input) size; wpk / wmeta = launch_pack_conv_weights_ws(w, C, 0).
// f32 = 1: exact fp32 MFMA form (wpk = launch_pack_conv_weights_ws32(w, C, 0); wmeta unused)
hipError_t launch_conv3x3_s2ws_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias, float* out,
                                     int N, int S, int c, int f32)
{
    IOD_XSKIP(512);
    if (!conv3x3_s2ws_ok(S, c) || !bias) return hipErrorInvalidValue;
    constexpr size_t lds = (size_t)2 * RW_BUFB + 32 + 64 * 4;
    static std::atomic<unsigned> attr_devs{0}, attr_devs32{0};
    if (hipError_t e = f32 ? iod_set_max_lds((const void*)conv3x3_s2ws_f16x3_kernel<64, true>, (int)lds, attr_devs32)
                           : iod_set_max_lds((const void*)conv3x3_s2ws_f16x3_kernel<64>, (int)lds, attr_devs); e != hipSuccess) return e;
    int n_cu = 0;
    if (hipError_t e = iod_cu_count(&n_cu); e != hipSuccess) return e;
    const int Sc = S / 2, tiles_x = (Sc + 15) / 16, tiles_y = (Sc + 1) / 2, ntiles = N * tiles_x * tiles_y;
    const int blocks = std::min(ntiles, 2 * n_cu);
    if (f32)
        hipLaunchKernelGGL((conv3x3_s2ws_f16x3_kernel<64, true>), dim3(blocks), dim3(256), lds, st, in, reinterpret_cast<const uint4*>(wpk), wmeta, bias,
                           out, Sc, tiles_x, tiles_y, ntiles);
    else
    hipLaunchKernelGGL((conv3x3_s2ws_f16x3_kernel<64>), dim3(blocks), dim3(256), lds, st, in, reinterpret_cast<const uint4*>(wpk), wmeta, bias,
                       out, Sc, tiles_x, tiles_y, ntiles);
    return hipGetLastError();
}
