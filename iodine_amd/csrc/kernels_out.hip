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
#include <cstdlib>
#include <type_traits>

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_ __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
}  // namespace

template <int C>
__global__ __launch_bounds__(256)
void dec_out_stream_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                                 const float* __restrict__ bias, float4* __restrict__ out, int S, int tiles, int ntiles)
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
    auto process = [&](int rb, auto nleft, f32x4 (&v)[NLD]) {
        constexpr int nl = decltype(nleft)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(nl) : "memory");
#pragma unroll
        for (int k = 0; k < NLD; ++k) asm volatile("" : "+v"(v[k]));
        __builtin_amdgcn_sched_barrier(0);
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w))));
        m = wave_max_f32(m);
        const float scale = tile_scale(m, 1.f);
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
        const float inv = inv_ws / scale;
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
    int t = blockIdx.x;
    set_tile(t);
    issue(rb_offset(wv), va);
    issue(rb_offset(wv + 4), vb);
    for (; t < ntiles; t += gridDim.x) {
        const int ctx = tx, cty = ty, cn = n;
        process(wv, integral_constant<int, NLD>{}, va);
        issue(rb_offset(wv + 8), va);
        process(wv + 4, integral_constant<int, NLD>{}, vb);
        process(wv + 8, integral_constant<int, 0>{}, va);
        if (t + (int)gridDim.x < ntiles) {                          // block-uniform
            set_tile(t + gridDim.x);
            issue(rb_offset(wv), va);
            issue(rb_offset(wv + 4), vb);
        }
        __syncthreads();
        const int y = tid >> 4, x = tid & 15;
        float4 o = b4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float4 q = *reinterpret_cast<const float4*>(s_P + ((y + tap / 3) * HALO + x + tap % 3) * PSTR + tap * 4);
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        out[((size_t)cn * S + cty * 16 + y) * S + ctx * 16 + x] = o;
        __syncthreads();                                            // P tile free for the next tile
    }
}

hipError_t launch_dec_out_stream_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                       const float* bias, float* out, int N, int S, int C)
{
    IOD_XSKIP(128);
    if (S % 16 != 0) return hipErrorInvalidValue;
    const int tiles = S / 16, ntiles = N * tiles * tiles;
    const int blocks = ntiles < 512 ? ntiles : 512;                 // two resident blocks per CU (63 KB LDS each), persistent
    if (C == 64) {
        constexpr size_t lds = (size_t)4 * 2 * 2 * 64 * 16 + 324 * 36 * 4;
        hipLaunchKernelGGL((dec_out_stream_f16x3_kernel<64>), dim3(blocks), dim3(256), lds, st, in,
                           reinterpret_cast<const uint4*>(wpk), wmeta, bias, reinterpret_cast<float4*>(out), S, tiles, ntiles);
    } else if (C == 32) {
        constexpr size_t lds = (size_t)2 * 2 * 2 * 64 * 16 + 324 * 36 * 4;
        hipLaunchKernelGGL((dec_out_stream_f16x3_kernel<32>), dim3(blocks), dim3(256), lds, st, in,
                           reinterpret_cast<const uint4*>(wpk), wmeta, bias, reinterpret_cast<float4*>(out), S, tiles, ntiles);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
