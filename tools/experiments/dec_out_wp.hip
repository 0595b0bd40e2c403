// EXPERIMENT (not built into the library): output conv forward with whole-pixel loads + a wave-private LDS transposition
// stage instead of fragment-shaped loads.  Drop-in for dec_out_stream_f16x3_kernel in iodine_amd/csrc/kernels_out.hip (same
// packs, same launch arguments).  Measured on MI355X, cfg3 (N = 224, 128 x 128, C = 64), same-process A/B in the training and
// the inference step: 308 us per launch against 266-268 us for the fragment-shaped loads (step 51.31 -> 51.50 ms, reconstruct
// 30.74 -> 30.94 ms): the eight offset computations, 16 LDS writes and 8 extra LDS reads per lane and row-block, and the three
// wave barriers per row-block cost more than the 4x fewer cache lines per load instruction save.  All 127 GPU tests passed
// with it as the default, so it is correct - just slower.
// =========================================================================================
// The same kernel with WHOLE-PIXEL loads (out_variant 2, default).  The fragment-shaped loads above touch 32 cache lines per
// instruction (16 bytes per lane at the pixel stride, two lanes per pixel): their issue alone takes 0.22 of the 0.27 ms.  Here
// a lane fetches 16 bytes of a pixel whose 256 (C = 64) bytes are covered by 16 consecutive lanes - 8 full lines per
// instruction - and the wave transposes its 32-pixel row-block into MFMA fragments through a private LDS stage:
// all channels' fp16 `hi` halves first (the passes W_lo.x_hi and W_hi.x_hi of every chunk), then the `lo` halves in the same
// 4.6 KB (W_hi.x_lo).  s_w + P tile + 4 stages = 81.5 KB: still two blocks per CU.
// =========================================================================================
template <int C>
__global__ __launch_bounds__(256, 2)
void dec_out_wp_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                             const float* __restrict__ bias, float4* __restrict__ out, int S, int tiles, int ntiles)
{
    constexpr int NCHUNK = C / 16;
    constexpr int HALO = 18, NPX = HALO * HALO;                     // 324 halo pixels = 11 row-blocks of 32 (last partial)
    constexpr int W_U4 = NCHUNK * 2 * 2 * 64;                       // [chunk][term hi/lo][kh][64 columns] uint4
    constexpr int PSTR = 36;                                        // floats per pixel in the P tile
    constexpr int SEGS = C / 4, PPI = 64 / SEGS, NLD = 32 / PPI;    // whole-pixel layout: 16-byte segments, pixels / instruction
    constexpr int STB = C * 2 + 16;                                 // bytes per staged pixel: C fp16 + pad (conflict-free b128 reads)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    uint4* s_w = reinterpret_cast<uint4*>(smem_b);
    float* s_P = reinterpret_cast<float*>(smem_b + W_U4 * 16);
    unsigned char* s_st = smem_b + W_U4 * 16 + NPX * PSTR * 4;      // [4 waves][32 px][STB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int seg = lane % SEGS, pl = lane / SEGS;
    for (int idx = tid; idx < W_U4; idx += 256) s_w[idx] = wpk[idx];
    const float inv_ws = wmeta[1];
    __syncthreads();                                                // (before the asm loads: hipcc's vmcnt(0) for the copy
                                                                    //  above would otherwise wait for them as well)
    unsigned char* st = s_st + wv * 32 * STB;
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
    // instruction k of row-block rb: halo pixel rb*32 + k*PPI + pl, 16-byte segment seg (outside the image / past pixel 323:
    // an offset beyond the descriptor's range, the hardware returns 0)
    auto issue = [&](int rb, f32x4 (&v)[NLD]) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int px = rb * 32 + k * PPI + pl;
            const int hy = (px * 3641) >> 16, hx = px - hy * HALO;  // px / 18 for px < 2048
            const int gy = ty * 16 - 1 + hy, gx = tx * 16 - 1 + hx;
            const bool ok = px < NPX && gy >= 0 && gy < S && gx >= 0 && gx < S;
            const unsigned off = ok ? (unsigned)(((gy * S + gx) * C + seg * 4) * 4) : 0x80000000u;
            asm volatile("s_nop 4" :: "s"(rsrc) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[k]) : "v"(off), "s"(rsrc) : "memory");
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
        // (the weight fragments are re-read from LDS per row-block: hoisted out of the tile loop they take 64 registers and spill)
        unsigned wofs = (unsigned)(kh * 64 + li);
        asm volatile("" : "+v"(wofs));
        // split: hi halves to the stage now, lo halves kept for the second round
        uint2 lo[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const f32x4 a = v[k] * scale;
            unsigned l0, l1;
            const unsigned h0 = pack_hi_lo(a.x, a.y, l0), h1 = pack_hi_lo(a.z, a.w, l1);
            lo[k] = make_uint2(l0, l1);
            *reinterpret_cast<uint2*>(st + (k * PPI + pl) * STB + seg * 8) = make_uint2(h0, h1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        f16x8 ax[NCHUNK];
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) ax[c] = *reinterpret_cast<const f16x8*>(st + li * STB + c * 32 + kh * 16);
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const uint4* q = s_w + (c * 2 + 0) * 2 * 64 + nt * 32 + wofs;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(q);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(q + 2 * 64);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ax[c], bl, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ax[c], bh, acc[nt], 0, 0, 0);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                            // every lane has its hi fragments: the stage takes the lo halves
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < NLD; ++k) *reinterpret_cast<uint2*>(st + (k * PPI + pl) * STB + seg * 8) = lo[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) ax[c] = *reinterpret_cast<const f16x8*>(st + li * STB + c * 32 + kh * 16);
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const uint4* q = s_w + (c * 2 + 0) * 2 * 64 + nt * 32 + wofs;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(q);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ax[c], bh, acc[nt], 0, 0, 0);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                            // stage free for the wave's next row-block
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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

    using std::integral_constant;
    const float4 b4 = make_float4(bias[0], bias[1], bias[2], bias[3]);
    f32x4 va[NLD], vb[NLD];
    int t = blockIdx.x;
    set_tile(t);
    issue(wv, va);
    issue(wv + 4, vb);
    for (; t < ntiles; t += gridDim.x) {
        const int ctx = tx, cty = ty, cn = n;
        process(wv, integral_constant<int, NLD>{}, va);
        issue(wv + 8, va);
        process(wv + 4, integral_constant<int, NLD>{}, vb);
        process(wv + 8, integral_constant<int, 0>{}, va);
        if (t + (int)gridDim.x < ntiles) {                          // block-uniform
            set_tile(t + gridDim.x);
            issue(wv, va);
            issue(wv + 4, vb);
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

hipError_t launch_dec_out_wp_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                   const float* bias, float* out, int N, int S, int C)
{
    IOD_XSKIP(128);
    if (S % 16 != 0) return hipErrorInvalidValue;
    const int tiles = S / 16, ntiles = N * tiles * tiles;
    const int blocks = ntiles < 512 ? ntiles : 512;                 // two resident blocks per CU, persistent
    if (C == 64) {
        constexpr size_t lds = (size_t)4 * 2 * 2 * 64 * 16 + 324 * 36 * 4 + 4 * 32 * (64 * 2 + 16);
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute((const void*)dec_out_wp_f16x3_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr = true;
        }
        hipLaunchKernelGGL((dec_out_wp_f16x3_kernel<64>), dim3(blocks), dim3(256), lds, st, in,
                           reinterpret_cast<const uint4*>(wpk), wmeta, bias, reinterpret_cast<float4*>(out), S, tiles, ntiles);
    } else if (C == 32) {
        constexpr size_t lds = (size_t)2 * 2 * 2 * 64 * 16 + 324 * 36 * 4 + 4 * 32 * (32 * 2 + 16);
        hipLaunchKernelGGL((dec_out_wp_f16x3_kernel<32>), dim3(blocks), dim3(256), lds, st, in,
                           reinterpret_cast<const uint4*>(wpk), wmeta, bias, reinterpret_cast<float4*>(out), S, tiles, ntiles);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

