// EXPERIMENT, not part of libiodine_hip.so: eight-wave form of the split-fp16 tile conv (round 1, "conv_variant=4").
// Same results bit for bit, same time; kept for reference only.
// Eight-wave form of the split-fp16 stride-1 tile conv (decoder 64->64 / 32->32 layers, lib/modeling/iodine.py:583,592
// forward and the autograd data gradient).  Same arithmetic, data layout, LDS image and per-accumulator MFMA order as
// conv3x3_tile_f16x3_kernel (kernels_conv.hip) - results are bitwise identical - but a 16x16 tile is worked on by 512
// threads: every wave owns 32 pixels x COUT channels instead of 64 x COUT.
//
// Why: the four-wave kernel needs 248 VGPRs (64 accumulators, two prefetch sets, double-buffered fragments), i.e. two
// waves per SIMD, and a block spends more than half of its life outside the MFMA phase (load wait, split + LDS writes,
// barriers, prefetch issue, epilogue: tools/tile_phase_prof.md), so two co-resident blocks keep the matrix pipe ~56 %
// busy.  Halving the per-wave tile halves every per-thread phase and the register budget (32 accumulators, one prefetch
// set, half-tap fragment buffers: <= 128 VGPRs), which puts FOUR waves on every SIMD at the same LDS footprint
// (2 blocks x 62.8 KB): while one wave stages or stores, three others can feed the pipe.  The price is 1.5x the LDS
// fragment traffic per MFMA (6 ds_read_b128 per 6 MFMAs instead of 8 per 12), still half of the pipe time.
#include "common.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_ __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));


}  // namespace

#ifdef IODINE_TILE_PROF
__device__ unsigned g_tile8_prof[TP_MAXBLK * 8];
#endif

template <int CIN, int COUT, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv3x3_tile8_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                                const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                                int S, int tiles, int rev)
{
    constexpr int NTHR = 512;
    constexpr int NCHUNK = CIN / 16;
    constexpr int NT = COUT / 32;
    constexpr int HALO = 18, NPX = HALO * HALO;
    constexpr int PXS = 80;                              // bytes per staged pixel: 32 hi + 32 lo + 16 pad
    constexpr int IN_BYTES = (NPX + 1) * PXS;            // +1 pixel: dump slot for idle lanes
    constexpr int W_U4 = 9 * 2 * 2 * COUT;               // uint4 (8 x fp16) per chunk
    constexpr int NIN = (NPX * 4 + NTHR - 1) / NTHR;     // float4 loads per thread per chunk (3)
    constexpr int NW = (W_U4 + NTHR - 1) / NTHR;         // uint4 loads per thread per chunk (5 / 3)
    static_assert(NCHUNK >= 1 && NCHUNK <= 4 && (NT == 1 || NT == 2), "unsupported channel counts");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* s_in = smem_b;
    uint4* s_w = reinterpret_cast<uint4*>(smem_b + IN_BYTES);
    float* s_max = reinterpret_cast<float*>(smem_b + IN_BYTES + W_U4 * 16);          // [8]

    TP_DECL;
    using std::integral_constant;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int prow = li >> 4, pcol = li & 15;

    int bid = rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;    // zig-zag launch order, see conv_f16x3()
    const int tx = bid % tiles; bid /= tiles;
    const int ty = bid % tiles;
    const int n = bid / tiles;

    // raw buffer loads / stores as inline asm with explicit counted waits: see conv3x3_tile_f16x3_kernel for the rules
    // (out-of-range offset 0x80000000 -> hardware zero, SGPR settle before VMEM, "+v" ties after the wait, s_nop after stores)
    auto make_rsrc = [&](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)base;
        i32x4_ r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
    const i32x4_ rsrc_in = make_rsrc(in + (size_t)n * S * S * CIN, (unsigned)(S * S * CIN * 4));
    const i32x4_ rsrc_w = make_rsrc(wpk, (unsigned)(NCHUNK * W_U4 * 16));
#define IOD_BLOAD4(dst, voff, rsrc, soff) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory")
#define IOD_SGPR_SETTLE(rsrc, soff) asm volatile("s_nop 4" :: "s"(rsrc), "s"(soff) : "memory")
    unsigned goff[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int idx = tid + k * NTHR;
        const int px = idx >> 2, cq = idx & 3;
        const int gy = ty * 16 - 1 + px / HALO, gx = tx * 16 - 1 + px % HALO;
        const bool ok = idx < NPX * 4 && gy >= 0 && gy < S && gx >= 0 && gx < S;
        goff[k] = ok ? (unsigned)(((gy * S + gx) * CIN + cq * 4) * 4) : 0x80000000u;
    }
    // packed weights: thread t copies uint4 t + k*512 of the chunk; the k*8 KiB step lives in the scalar offset, only the
    // last (partial) round needs its own masked offset
    const unsigned woff = (unsigned)(tid * 16);
    const unsigned woff_last = tid + (NW - 1) * NTHR < W_U4 ? woff : 0x80000000u;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // ONE prefetch set: the next chunk's input and weights are requested right after the current chunk has been staged
    // and land under its MFMA phase (the other three waves of the SIMD cover the first chunk's exposed latency)
    f32x4 rin[NIN];
    u32x4_ rw[NW];
    auto prefetch = [&](int chunk) {
        const int soff = chunk * 64;                         // 16 channels
        IOD_SGPR_SETTLE(rsrc_in, soff);
#pragma unroll
        for (int k = 0; k < NIN; ++k) IOD_BLOAD4(rin[k], goff[k], rsrc_in, soff);
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int woffs = chunk * (W_U4 * 16) + k * (NTHR * 16);
            IOD_SGPR_SETTLE(rsrc_w, woffs);
            if (k < NW - 1) IOD_BLOAD4(rw[k], woff, rsrc_w, woffs);
            else IOD_BLOAD4(rw[k], woff_last, rsrc_w, woffs);
        }
    };
    auto vm_wait_all = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NIN; ++k) asm volatile("" : "+v"(rin[k]));
#pragma unroll
        for (int k = 0; k < NW; ++k) asm volatile("" : "+v"(rw[k]));
        __builtin_amdgcn_sched_barrier(0);
    };
    float cur_scale = 1.f;
    auto commit = [&]() -> float {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const f32x4 v = rin[k];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        m = wave_max_f32(m);
        TP_STAMP(1);                                         // [1] wait for the chunk's global loads + max
        if (lane == 0) s_max[wv] = m;
        __syncthreads();                                     // also: every wave is done reading the previous chunk
        TP_STAMP(2);                                         // [2] barrier 1
        const float4 ma = *reinterpret_cast<const float4*>(s_max), mc = *reinterpret_cast<const float4*>(s_max + 4);
        const float mb = fmaxf(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w)), fmaxf(fmaxf(mc.x, mc.y), fmaxf(mc.z, mc.w)));
        const float scale = tile_scale(mb, cur_scale);
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int idx = tid + k * NTHR;
            const int px = idx < NPX * 4 ? idx >> 2 : NPX, cq = idx & 3;          // idle lanes write the dump slot
            f32x4 v = rin[k];
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            const float hx = __uint_as_float(__float_as_uint(v.x) & 0xffffe000u), hy = __uint_as_float(__float_as_uint(v.y) & 0xffffe000u);
            const float hz = __uint_as_float(__float_as_uint(v.z) & 0xffffe000u), hw = __uint_as_float(__float_as_uint(v.w) & 0xffffe000u);
            typedef __fp16 h2 __attribute__((ext_vector_type(2)));
            const h2 h01 = __builtin_amdgcn_cvt_pkrtz(hx, hy), h23 = __builtin_amdgcn_cvt_pkrtz(hz, hw);
            const h2 l01 = __builtin_amdgcn_cvt_pkrtz(v.x - hx, v.y - hy), l23 = __builtin_amdgcn_cvt_pkrtz(v.z - hz, v.w - hw);
            uint2 hi, lo;
            __builtin_memcpy(&hi.x, &h01, 4); __builtin_memcpy(&hi.y, &h23, 4);
            __builtin_memcpy(&lo.x, &l01, 4); __builtin_memcpy(&lo.y, &l23, 4);
            *reinterpret_cast<uint2*>(s_in + px * PXS + cq * 8) = hi;
            *reinterpret_cast<uint2*>(s_in + px * PXS + 32 + cq * 8) = lo;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int idx = tid + k * NTHR;
            if (idx < W_U4) s_w[idx] = make_uint4(rw[k].x, rw[k].y, rw[k].z, rw[k].w);
        }
        TP_STAMP(3);                                         // [3] scale, split, LDS writes
        __syncthreads();
        TP_STAMP(4);                                         // [4] barrier 2
        return scale;
    };
    auto rescale = [&](float new_scale) {
        if (new_scale != cur_scale) {                        // block-uniform
            const float r = new_scale / cur_scale;           // exact: both are powers of two
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[nt][q] *= r;
            cur_scale = new_scale;
        }
    };

    // Fragments: FA = the wave's 32 pixels at one tap (hi, lo), FB = 32 output channels at one (tap, channel half).  Both
    // ping-pong; the unit of software pipelining is a HALF tap (3 MFMAs on one accumulator): issue the next half-step's
    // reads, wait (counted, LDS returns in order) for the current one's, MFMA.
    struct FA { f16x8 h, l; };
    struct FB { f16x8 h, l; };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_b;
    const unsigned a_addr = lds_base + ((2 * wv + prow) * HALO + pcol) * PXS + kh * 16;
    const unsigned b_addr = lds_base + IN_BYTES + (kh * COUT + li) * 16;
#define IOD_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto LOADA = [&, a_addr](auto tapc, FA& f) {
        constexpr int tap = decltype(tapc)::value;
        constexpr int aoff = ((tap / 3) * HALO + (tap % 3)) * PXS;
        const unsigned aa = a_addr;
        IOD_DSR128(f.h, aa, aoff);
        IOD_DSR128(f.l, aa, aoff + 32);
    };
    auto LOADB = [&, b_addr](auto hc, FB& f) {
        constexpr int h = decltype(hc)::value;
        constexpr int tap = h / NT, nt = h % NT;
        constexpr int boff = tap * 4 * COUT * 16 + nt * 512;
        const unsigned ba = b_addr;
        IOD_DSR128(f.h, ba, boff);
        IOD_DSR128(f.l, ba, boff + 2 * COUT * 16);
    };
#undef IOD_DSR128
    constexpr int NH = 9 * NT;                               // half-steps per chunk
    FA fa[2];
    FB fb[2];
#define IOD_HSTEP(H)                                                                                          \
    {                                                                                                         \
        constexpr int h_ = (H), tap_ = h_ / NT, nt_ = h_ % NT;                                                \
        if constexpr (h_ + 1 < NH) {                                                                          \
            LOADB(integral_constant<int, (h_ + 1 < NH ? h_ + 1 : 0)>{}, fb[(h_ + 1) & 1]);                    \
            if constexpr ((h_ + 1) % NT == 0) {                                                               \
                LOADA(integral_constant<int, (tap_ + 1 < 9 ? tap_ + 1 : 0)>{}, fa[(tap_ + 1) & 1]);           \
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                            \
            } else {                                                                                          \
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                                            \
            }                                                                                                 \
        } else {                                                                                              \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        acc[nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[h_ & 1].h, fa[tap_ & 1].l, acc[nt_], 0, 0, 0);   \
        acc[nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[h_ & 1].l, fa[tap_ & 1].h, acc[nt_], 0, 0, 0);   \
        acc[nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[h_ & 1].h, fa[tap_ & 1].h, acc[nt_], 0, 0, 0);   \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    auto compute = [&]() {
        TP_STAMP(5);                                         // [5] issue of the next prefetch (between commit and compute)
        LOADA(integral_constant<int, 0>{}, fa[0]);
        LOADB(integral_constant<int, 0>{}, fb[0]);
        IOD_HSTEP(0) IOD_HSTEP(1) IOD_HSTEP(2) IOD_HSTEP(3) IOD_HSTEP(4) IOD_HSTEP(5) IOD_HSTEP(6) IOD_HSTEP(7) IOD_HSTEP(8)
        if constexpr (NT == 2) {
            IOD_HSTEP(9) IOD_HSTEP(10) IOD_HSTEP(11) IOD_HSTEP(12) IOD_HSTEP(13) IOD_HSTEP(14) IOD_HSTEP(15) IOD_HSTEP(16)
            IOD_HSTEP(17)
        }
        TP_STAMP(6);                                         // [6] 9 taps of LDS fragment reads + MFMA
    };

    // data-gradient form: the ELU' operand is requested after the LAST chunk has been staged (see the four-wave kernel)
    f32x4 ax[NT][4];
    const unsigned voff = (unsigned)((((ty * 16 + 2 * wv + prow) * S + tx * 16 + pcol) * COUT + 4 * kh) * 4);
    auto prefetch_aux = [&]() {
        if constexpr (EPI == EPI_MUL_ELUGRAD) {
            const i32x4_ rsrc_aux = make_rsrc(aux + (size_t)n * S * S * COUT, (unsigned)(S * S * COUT * 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int soff = (nt * 32 + 8 * g4) * 4;
                    IOD_SGPR_SETTLE(rsrc_aux, soff);
                    IOD_BLOAD4(ax[nt][g4], voff, rsrc_aux, soff);
                }
        }
    };
    TP_STAMP(0);                                             // [0] block start: index arithmetic
    prefetch(0);
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        vm_wait_all();
        const float sc = commit();
        if (c == 0) cur_scale = sc; else rescale(sc);
        if (c + 1 < NCHUNK) prefetch(c + 1); else prefetch_aux();
        compute();
    }
#undef IOD_HSTEP

    const float inv_ws = wmeta[1] / cur_scale;
    static_assert(EPI == EPI_BIAS_ELU || EPI == EPI_MUL_ELUGRAD, "the C -> 4 output conv has its own GEMM-form kernel");
    {
        const i32x4_ rsrc_out = make_rsrc(out + (size_t)n * S * S * COUT, (unsigned)(S * S * COUT * 4));
        f32x4 bv[NT][4];
        if constexpr (EPI == EPI_BIAS_ELU) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 t = *reinterpret_cast<const float4*>(bias + nt * 32 + 8 * g4 + 4 * kh);
                    bv[nt][g4] = f32x4{t.x, t.y, t.z, t.w};
                }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(bv[nt][g4]));
        }
        if constexpr (EPI == EPI_MUL_ELUGRAD) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(ax[nt][g4]));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v = f32x4{acc[nt][4 * g4] * inv_ws, acc[nt][4 * g4 + 1] * inv_ws,
                                acc[nt][4 * g4 + 2] * inv_ws, acc[nt][4 * g4 + 3] * inv_ws};
                if constexpr (EPI == EPI_BIAS_ELU) {
                    const f32x4 b4 = bv[nt][g4];
                    v = f32x4{elu1_fast(v.x + b4.x), elu1_fast(v.y + b4.y), elu1_fast(v.z + b4.z), elu1_fast(v.w + b4.w)};
                } else if constexpr (EPI == EPI_MUL_ELUGRAD) {
                    const f32x4 a4 = ax[nt][g4];
                    v.x *= elu1_grad_from_out(a4.x); v.y *= elu1_grad_from_out(a4.y);
                    v.z *= elu1_grad_from_out(a4.z); v.w *= elu1_grad_from_out(a4.w);
                }
                const int soff = (nt * 32 + 8 * g4) * 4;
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rsrc_out), "s"(soff) : "memory");
            }
    }
    TP_STAMP(7);                                             // [7] epilogue (issue of the stores)
    TP_FLUSH(g_tile8_prof);
#undef IOD_BLOAD4
#undef IOD_SGPR_SETTLE
}

template <int CIN, int COUT, int EPI>
static hipError_t launch_tile8_inst(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                    const float* bias, const float* aux, float* out, int N, int S, int rev)
{
    constexpr size_t lds = (size_t)(18 * 18 + 1) * 80 + (size_t)9 * 2 * 2 * COUT * 16 + 32;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_tile8_f16x3_kernel<CIN, COUT, EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles = S / 16;
    hipLaunchKernelGGL((conv3x3_tile8_f16x3_kernel<CIN, COUT, EPI>), dim3(N * tiles * tiles), dim3(512), lds, st, in,
                       reinterpret_cast<const uint4*>(wpk), wmeta, bias, aux, out, S, tiles, rev);
#ifdef IODINE_TILE_PROF
    {
        const int nb = std::min(N * tiles * tiles, TP_MAXBLK);
        std::vector<unsigned> hp((size_t)nb * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_tile8_prof), hp.size() * sizeof(unsigned));
        static const char* names[8] = {"start", "load-wait+max", "barrier1", "split+lds-write", "barrier2", "prefetch-issue",
                                       "taps(lds-read+mfma)", "epilogue"};
        double sum[8] = {0}, tot = 0;
        for (int b2 = 0; b2 < nb; ++b2) for (int i = 0; i < 8; ++i) sum[i] += hp[(size_t)b2 * 8 + i];
        for (int i = 0; i < 8; ++i) tot += sum[i] / nb;
        fprintf(stderr, "[tile8 prof] memtime ticks per block (wave 0), total %.0f:", tot);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f |", names[i], sum[i] / nb);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

hipError_t launch_conv3x3_tile8_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                      const float* bias, const float* aux, float* out, int N, int S, int cin, int cout,
                                      int epi, int rev)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
#define T8_CASE(CI, CO, EP) \
    if (cin == CI && cout == CO && epi == EP) return launch_tile8_inst<CI, CO, EP>(st, in, wpk, wmeta, bias, aux, out, N, S, rev);
    T8_CASE(64, 64, EPI_BIAS_ELU) T8_CASE(64, 64, EPI_MUL_ELUGRAD)
    T8_CASE(32, 32, EPI_BIAS_ELU) T8_CASE(32, 32, EPI_MUL_ELUGRAD)
#undef T8_CASE
    return hipErrorInvalidValue;
}
