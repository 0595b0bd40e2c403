// EXPERIMENT, not part of libiodine_hip.so: the warp-specialised persistent form of the split-fp16 tile conv (round 1,
// "conv_variant=3").  Same results as the product kernel bit for bit, same time.  Needs the helpers of
// iodine_amd/csrc/kernels_conv.hip above its original position to compile; kept for reference only.
// =========================================================================================
// v3 of the split-fp16 tile conv: warp-specialised persistent kernel.  512 threads, one block per CU:
//   waves 0-3  CONSUMERS: LDS fragment reads + MFMA only (64 px x COUT per wave), epilogue stores
//   waves 4-7  PRODUCERS: global loads (input halo chunk 3 steps ahead, packed weights 2 steps ahead), power-of-two
//              scaling, fp32 -> fp16 hi/lo split, LDS writes into the OTHER buffer, tile-chunk max |x|
// One raw s_barrier per (tile, chunk) step; no LDS-DMA (so hipcc keeps counted vmcnt waits) and no conditional
// stages (steps past the end are clamped and land in buffers nobody reads).  The measured serialisation of the
// one-tile-per-block kernel (load 0.25 + convert 0.24 + MFMA 0.49 + store 0.13 ms per 64->64 layer at cfg3) is
// what this removes: producers and consumers are different waves on the same SIMDs, MFMA and VALU/VMEM co-issue.
// =========================================================================================
template <int CIN, int COUT, int EPI>
__global__ __launch_bounds__(512, 2)
void conv3x3_tile_f16x3_v3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk,
                                  const float* __restrict__ wmeta, const float* __restrict__ bias,
                                  const float* __restrict__ aux, float* __restrict__ out, int S, int tiles, int ntiles,
                                  int flags, unsigned long long* __restrict__ prof)
{
    constexpr int NCHUNK = CIN / 16;
    constexpr int NT = COUT / 32;
    constexpr int HALO = 18, PXS = 80, NPX = HALO * HALO, IN_BYTES = (NPX + 1) * PXS;
    constexpr int W_U4 = 9 * 2 * 2 * COUT, W_BYTES = W_U4 * 16;
    constexpr int NPT = 192;                               // input-producer threads (waves 4-6); wave 7 streams weights
    constexpr int NIN = (NPX * 4 + NPT - 1) / NPT;         // float4 per input-producer thread per chunk (7)
    constexpr int W_PIECES = W_U4 / 64;                    // 1 KiB LDS-DMA pieces per chunk (36 / 18)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    float* s_max = reinterpret_cast<float*>(smem_b + 2 * IN_BYTES + 3 * W_BYTES);        // [3][4]
    // weights are TRIPLE buffered (slot = step % 3) so that a chunk's weights are fetched and stored within one
    // producer iteration (no register array living across barriers)

    const int tid = threadIdx.x;
    const int lane = tid & 63, kh = lane >> 5, li = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int ptid = tid - 256;                             // input-producer thread index (0..191)
    const int cwv = wv & 3;                                 // consumer wave: tile rows 4*cwv .. +3

    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nq = my_tiles * NCHUNK;
    if (nq == 0) return;

    auto tile_coords = [&](int q, int& n, int& ty, int& tx) {
        int t = blockIdx.x + (q / NCHUNK) * gridDim.x;
        tx = t % tiles; t /= tiles;
        ty = t % tiles; n = t / tiles;
    };
    auto SCALE = [&](int slot) -> float {
        if (flags & 8) return 1.f;
        const float* p = s_max + slot * 4;
        const float mb = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
        const int e = (int)((__float_as_uint(mb) >> 23) & 0xffu) - 127;
        int se = 12 - e;
        se = se > 100 ? 100 : (se < -100 ? -100 : se);
        return (mb > 0.f && mb < 3.0e38f) ? __uint_as_float((unsigned)(127 + se) << 23) : 1.f;
    };
    unsigned long long t_wait = 0, t_last = __builtin_amdgcn_s_memtime(), t_work = 0;
    auto block_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes / reads retired
        if (prof) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            t_work += t0 - t_last; t_wait += t1 - t0; t_last = t1;
        } else {
            __builtin_amdgcn_s_barrier();
        }
    };
    auto prof_flush = [&](int role) {
        if (prof && lane == 0) { atomicAdd(&prof[role * 2], t_work); atomicAdd(&prof[role * 2 + 1], t_wait); }
    };

    if (producer && wv == 7) {
        // ------------------------------------------------------------ WEIGHT STREAMER ----
        // packed fp16 weights go global -> LDS by DMA (no VGPR round trip); this wave issues nothing else, so its
        // vmcnt(0) before each barrier only covers its own pieces.
        auto DMA = [&](int q_, int wslot) {
            const int q = q_ < nq ? q_ : nq - 1;
            const uint4* wsrc = wpk + (size_t)(q % NCHUNK) * W_U4;
            unsigned char* wdst = smem_b + 2 * IN_BYTES + wslot * W_BYTES;
#pragma unroll 4
            for (int piece = 0; piece < W_PIECES; ++piece)
                if (!(flags & 4)) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(wdst + piece * 1024), 16, 0, 0);
        };
        auto wbarrier = [&]() {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (prof) {
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                t_work += t0 - t_last; t_wait += t1 - t0; t_last = t1;
            } else {
                __builtin_amdgcn_s_barrier();
            }
        };
        DMA(0, 0);
        DMA(1, 1);
        if (lane == 0) { s_max[3] = 0.f; s_max[7] = 0.f; s_max[11] = 0.f; }      // unused 4th producer slot
        wbarrier();                                         // P1
        wbarrier();                                         // P2
        for (int q = 0; q < nq; ++q) {
            DMA(q + 2, (q + 2) % 3);
            wbarrier();
        }
        prof_flush(2);
        return;
    }
    if (producer) {
        // ------------------------------------------------------------ INPUT PRODUCERS ----
        // per-thread element offsets of its NIN halo float4s inside the current look-ahead tile (32-bit: tensors are
        // < 2^31 floats), recomputed only when the look-ahead step enters a new tile; -1 = outside the image / idle
        // raw buffer loads (inline asm, explicit counted vmcnt): one descriptor per slot-image, halo pixels outside the
        // image / idle lanes carry an out-of-range offset and come back as 0, the chunk offset is the scalar operand
        typedef int i32x4_ __attribute__((ext_vector_type(4)));
        auto make_rsrc = [&](const void* base, unsigned bytes) {
            const unsigned long long p = (unsigned long long)base;
            i32x4_ r;
            r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
            r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
            r.z = __builtin_amdgcn_readfirstlane((int)bytes);
            r.w = 0x00020000;
            return r;
        };
        unsigned goff[NIN];
        i32x4_ rsrc_in = make_rsrc(in, 0u);
        auto tile_offsets = [&](int q) {
            int n, ty, tx;
            tile_coords(q, n, ty, tx);
            const int y0 = ty * 16 - 1, x0 = tx * 16 - 1;
            rsrc_in = make_rsrc(in + (size_t)n * S * S * CIN, (unsigned)(S * S * CIN * 4));
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const int idx = ptid + k * NPT;
                const int px = idx >> 2, cq = idx & 3;
                const int gy = y0 + px / HALO, gx = x0 + px % HALO;
                const bool ok = idx < NPX * 4 && gy >= 0 && gy < S && gx >= 0 && gx < S;
                goff[k] = ok ? (unsigned)(((gy * S + gx) * CIN + cq * 4) * 4) : 0x80000000u;
            }
        };
        auto G = [&](int q_, f32x4 (&r)[NIN]) {
            const int q = q_ < nq ? q_ : nq - 1;
            if (q % NCHUNK == 0 || q_ <= 2) tile_offsets(q);            // block-uniform
            const int soff = (q % NCHUNK) * 64;
            asm volatile("s_nop 4" :: "s"(rsrc_in), "s"(soff) : "memory");
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[k]) : "v"(goff[k]), "s"(rsrc_in), "s"(soff) : "memory");
        };
        auto wait_set = [&](auto nc, f32x4 (&r)[NIN]) {
            constexpr int nleft = decltype(nc)::value;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(nleft) : "memory");
#pragma unroll
            for (int k = 0; k < NIN; ++k) asm volatile("" : "+v"(r[k]));
            __builtin_amdgcn_sched_barrier(0);
        };
        auto MAXPUB = [&](int slot, const f32x4 (&r)[NIN]) {
            if (flags & 8) return;
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(r[k].x), fabsf(r[k].y)), fmaxf(fabsf(r[k].z), fabsf(r[k].w))));
            m = wave_max_f32(m);
            if (lane == 0) s_max[slot * 4 + (wv - 4)] = m;
        };
        auto WIN = [&](int q, const f32x4 (&r)[NIN], float scale) {
            if (flags & 32) return;
            unsigned char* s_in = smem_b + (q & 1) * IN_BYTES;
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const int idx = ptid + k * NPT;
                const int px = idx < NPX * 4 ? idx >> 2 : NPX, cq = idx & 3;      // idle lanes write the dump slot
                f32x4 v = r[k];
                v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                // hi = v truncated to fp16 precision (mask the 13 low mantissa bits: exact in fp16 for the scaled
                // range), lo = v - hi (exact in fp32), both packed with v_cvt_pkrtz (2 values per instruction)
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
        };

        f32x4 R0[NIN], R1[NIN], R2[NIN];
        using std::integral_constant;
        // prologue: stage chunk 0, chunks 1 and 2 in flight, max(1) published
        G(0, R0);
        G(1, R1); G(2, R2);
        wait_set(integral_constant<int, 2 * NIN>{}, R0);
        MAXPUB(0, R0);
        block_barrier();                                    // P1: max(0) visible
        WIN(0, R0, SCALE(0));
        wait_set(integral_constant<int, NIN>{}, R1);
        MAXPUB(1, R1);
        block_barrier();                                    // P2: chunk 0 staged, max(1) visible

        // iteration q: fetch input q+3, stage input q+1, publish max(q+2)
        auto iteration = [&](int q, int sl, f32x4 (&Rnext)[NIN], f32x4 (&Rmax)[NIN], f32x4 (&Rload)[NIN]) {
            G(q + 3, Rload);
            wait_set(integral_constant<int, 2 * NIN>{}, Rnext);          // (arrived during the previous step)
            WIN(q + 1, Rnext, SCALE((sl + 1) % 3));
            wait_set(integral_constant<int, NIN>{}, Rmax);               // requested two steps ago
            MAXPUB((sl + 2) % 3, Rmax);
            block_barrier();
        };
        for (int q = 0; q < nq; q += 3) {
            iteration(q, 0, R1, R2, R0);
            if (q + 1 < nq) iteration(q + 1, 1, R2, R0, R1);
            if (q + 2 < nq) iteration(q + 2, 2, R0, R1, R2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        prof_flush(1);
        return;
    }

    // ---------------------------------------------------------------------- CONSUMER ----
    // The step loop is software-pipelined ACROSS the barrier: once the fragments of a step's last tap are in registers the
    // wave has no LDS reads of that step left, so it arrives at the barrier first, requests tap 0 of the NEXT step (and the
    // next scale) behind it, and only then issues the last tap's 12 MFMAs - barrier skew, the scale read and the LDS
    // latency of the first fragments hide under them instead of leaving the matrix pipe idle at every step boundary.
    // Nine taps per step alternate the two fragment sets, so consecutive steps start on opposite sets (loop unrolled by 2).
    const int prow = li >> 4, pcol = li & 15;
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const float inv_ws = wmeta[1];
    block_barrier();                                        // P1
    float cur_scale = SCALE(0);
    block_barrier();                                        // P2
    float sc_next = SCALE(1);                               // scale of chunk 1 (published before P2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    struct Frag { f16x8 ah[2], al[2], bh[NT], bl[NT]; };
    constexpr int NRD = 4 + 2 * NT;                         // ds_read_b128 per tap
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_b;
    const unsigned a_lane = lds_base + ((4 * cwv + prow) * HALO + pcol) * PXS + kh * 16;
    const unsigned b_lane = lds_base + 2 * IN_BYTES + (kh * COUT + li) * 16;
    const unsigned m_addr = lds_base + 2 * IN_BYTES + 3 * W_BYTES;                       // s_max[3][4]
#define IOD_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    // one fragment read of tap `tapc` (i = 0 .. NRD-1) / one MFMA of a tap (i = 0 .. NMF-1, same order as the MMA of the
    // one-tile-per-block kernel: results are bitwise identical)
    constexpr int NMF = 6 * NT;
    auto RD = [](auto ic, auto tapc, Frag& f, unsigned a_addr0, unsigned b_addr) {
        constexpr int i = decltype(ic)::value, tap = decltype(tapc)::value;
        constexpr int aoff = ((tap / 3) * HALO + (tap % 3)) * PXS;
        constexpr int boff = tap * 4 * COUT * 16;
        const unsigned a_addr1 = a_addr0 + 2 * HALO * PXS;
        if constexpr (i == 0) IOD_DSR128(f.ah[0], a_addr0, aoff);
        else if constexpr (i == 1) IOD_DSR128(f.al[0], a_addr0, aoff + 32);
        else if constexpr (i == 2) IOD_DSR128(f.ah[1], a_addr1, aoff);
        else if constexpr (i == 3) IOD_DSR128(f.al[1], a_addr1, aoff + 32);
        else if constexpr (i == 4) IOD_DSR128(f.bh[0], b_addr, boff);
        else if constexpr (i == 5) IOD_DSR128(f.bl[0], b_addr, boff + 2 * COUT * 16);
        else if constexpr (i == 6 && NT == 2) IOD_DSR128(f.bh[NT - 1], b_addr, boff + 512);
        else if constexpr (i == 7 && NT == 2) IOD_DSR128(f.bl[NT - 1], b_addr, boff + 2 * COUT * 16 + 512);
    };
    auto MF = [&](auto ic, const Frag& f) {
        constexpr int i = decltype(ic)::value, term = i / (2 * NT), mt = (i % (2 * NT)) / NT, nt = i % NT;
        if constexpr (term == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[nt], f.al[mt], acc[mt][nt], 0, 0, 0);
        else if constexpr (term == 1) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[nt], f.ah[mt], acc[mt][nt], 0, 0, 0);
        else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[nt], f.ah[mt], acc[mt][nt], 0, 0, 0);
    };
    using std::integral_constant;
    // a tap: its MFMAs with the NEXT tap's fragment reads slotted in behind the first NRD of them (one LDS instruction per
    // MFMA gap: the matrix pipe never waits for a block of eight reads to issue)
    auto TAP = [&](const Frag& fc, Frag& fn, auto tapn, unsigned a_n, unsigned b_n) {
#define IOD_MR(I)                                                                                              \
        if constexpr (I < NMF) { MF(integral_constant<int, (I < NMF ? I : 0)>{}, fc); __builtin_amdgcn_sched_barrier(0); }       \
        if constexpr (I < NRD) { RD(integral_constant<int, I>{}, tapn, fn, a_n, b_n); __builtin_amdgcn_sched_barrier(0); }
        IOD_MR(0) IOD_MR(1) IOD_MR(2) IOD_MR(3) IOD_MR(4) IOD_MR(5) IOD_MR(6) IOD_MR(7) IOD_MR(8) IOD_MR(9) IOD_MR(10) IOD_MR(11)
#undef IOD_MR
    };
#define IOD_STEP(T, FCUR, FNEXT)                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        TAP(FCUR, FNEXT, integral_constant<int, T + 1>{}, a0, bb);
    auto step_body = [&](int q, Frag& fa, Frag& fb) {
        const int sl = q % 3;
        const unsigned a0 = a_lane + (q & 1) * IN_BYTES, bb = b_lane + sl * W_BYTES;
        IOD_STEP(0, fa, fb) IOD_STEP(1, fb, fa) IOD_STEP(2, fa, fb) IOD_STEP(3, fb, fa)
        IOD_STEP(4, fa, fb) IOD_STEP(5, fb, fa) IOD_STEP(6, fa, fb) IOD_STEP(7, fb, fa)
        // tap 8 (fragments in fa): every LDS read of this step has been issued; block_barrier() retires them and arrives
        block_barrier();
        f32x4 mraw;                                          // max |x| of chunk q + 2 (published during this step)
        {
            const unsigned ma = m_addr + ((sl + 2) % 3) * 16;
            asm volatile("ds_read_b128 %0, %1" : "=v"(mraw) : "v"(ma));
        }
        __builtin_amdgcn_sched_barrier(0);
        TAP(fa, fb, integral_constant<int, 0>{}, a_lane + ((q + 1) & 1) * IN_BYTES, b_lane + ((q + 1) % 3) * W_BYTES);
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NRD) : "memory");      // the scale read (older than the fragment reads)
        asm volatile("" : "+v"(mraw));
        __builtin_amdgcn_sched_barrier(0);
        float sc_next2 = 1.f;
        if (!(flags & 8)) {
            const float mb = fmaxf(fmaxf(mraw.x, mraw.y), fmaxf(mraw.z, mraw.w));
            const int e = (int)((__float_as_uint(mb) >> 23) & 0xffu) - 127;
            int se = 12 - e;
            se = se > 100 ? 100 : (se < -100 ? -100 : se);
            sc_next2 = (mb > 0.f && mb < 3.0e38f) ? __uint_as_float((unsigned)(127 + se) << 23) : 1.f;
        }
        if (q % NCHUNK == NCHUNK - 1) {
            int n, ty, tx;
            tile_coords(q, n, ty, tx);
            const float inv = inv_ws / cur_scale;
            // accumulator rows are channels (MFMAs issued as (weights, activations)): float4 buffer stores, see the
            // one-tile-per-block kernel above
            typedef int i32x4_ __attribute__((ext_vector_type(4)));
            auto make_rsrc = [&](const void* base, unsigned bytes) {
                const unsigned long long p = (unsigned long long)base;
                i32x4_ r;
                r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
                r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
                r.z = __builtin_amdgcn_readfirstlane((int)bytes);
                r.w = 0x00020000;
                return r;
            };
            if (!(flags & 16)) {                                 // (timing experiment, conv_ablate: 16 = skip the epilogue)
                const i32x4_ rsrc_out = make_rsrc(out + (size_t)n * S * S * COUT, (unsigned)(S * S * COUT * 4));
                unsigned voff[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int gy = ty * 16 + 4 * cwv + 2 * mt + (li >> 4), gx = tx * 16 + (li & 15);
                    voff[mt] = (unsigned)(((gy * S + gx) * COUT + 4 * kh) * 4);
                }
                f32x4 bv[NT][4], ax[2][NT][4];
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
                    const i32x4_ rsrc_aux = make_rsrc(aux + (size_t)n * S * S * COUT, (unsigned)(S * S * COUT * 4));
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const int soff = (nt * 32 + 8 * g4) * 4;
                                asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen"
                                             : "=v"(ax[mt][nt][g4]) : "v"(voff[mt]), "s"(rsrc_aux), "s"(soff) : "memory");
                            }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(ax[mt][nt][g4]));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            f32x4 v = f32x4{acc[mt][nt][4 * g4] * inv, acc[mt][nt][4 * g4 + 1] * inv,
                                            acc[mt][nt][4 * g4 + 2] * inv, acc[mt][nt][4 * g4 + 3] * inv};
                            if constexpr (EPI == EPI_BIAS_ELU) {
                                const f32x4 b4 = bv[nt][g4];
                                v = f32x4{elu1_fast(v.x + b4.x), elu1_fast(v.y + b4.y), elu1_fast(v.z + b4.z), elu1_fast(v.w + b4.w)};
                            } else if constexpr (EPI == EPI_MUL_ELUGRAD) {
                                const f32x4 a4 = ax[mt][nt][g4];
                                v.x *= elu1_grad_from_out(a4.x); v.y *= elu1_grad_from_out(a4.y);
                                v.z *= elu1_grad_from_out(a4.z); v.w *= elu1_grad_from_out(a4.w);
                            }
                            const int soff = (nt * 32 + 8 * g4) * 4;
                            asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1"
                                         :: "v"(v), "v"(voff[mt]), "s"(rsrc_out), "s"(soff) : "memory");
                        }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            cur_scale = sc_next;
        } else if (sc_next != cur_scale) {
            const float r = sc_next / cur_scale;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mt][nt][e] *= r;
            cur_scale = sc_next;
        }
        sc_next = sc_next2;
    };
#undef IOD_STEP
    Frag f0, f1;
    {
        using Z = integral_constant<int, 0>;
        RD(integral_constant<int, 0>{}, Z{}, f0, a_lane, b_lane); RD(integral_constant<int, 1>{}, Z{}, f0, a_lane, b_lane);
        RD(integral_constant<int, 2>{}, Z{}, f0, a_lane, b_lane); RD(integral_constant<int, 3>{}, Z{}, f0, a_lane, b_lane);
        RD(integral_constant<int, 4>{}, Z{}, f0, a_lane, b_lane); RD(integral_constant<int, 5>{}, Z{}, f0, a_lane, b_lane);
        RD(integral_constant<int, 6>{}, Z{}, f0, a_lane, b_lane); RD(integral_constant<int, 7>{}, Z{}, f0, a_lane, b_lane);
    }
    for (int q = 0; q < nq; q += 2) {                       // nq = tiles * NCHUNK is even
        step_body(q, f0, f1);
        step_body(q + 1, f1, f0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef IOD_DSR128
    prof_flush(0);
}

template <int CIN, int COUT, int EPI>
static hipError_t launch_tile_f16x3_v3_inst(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                            const float* bias, const float* aux, float* out, int N, int S)
{
    constexpr size_t lds = (size_t)2 * (18 * 18 + 1) * 80 + (size_t)3 * 9 * 2 * 2 * COUT * 16 + 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_tile_f16x3_v3_kernel<CIN, COUT, EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles = S / 16, ntiles = N * tiles * tiles;
    const int blocks = ntiles < 256 ? ntiles : 256;
    static int flags = -1;
    if (flags < 0) { const char* e = getenv("IODINE_CONV_ABLATE"); flags = e ? atoi(e) : 0; }
    unsigned long long* prof = nullptr;
    if (getenv("IODINE_CONV_PROF")) {                       // debug: per-role work / barrier-wait cycle totals
        static unsigned long long* dprof = nullptr;
        if (!dprof) (void)hipMalloc((void**)&dprof, 64);
        (void)hipMemsetAsync(dprof, 0, 64, st);
        prof = dprof;
    }
    hipLaunchKernelGGL((conv3x3_tile_f16x3_v3_kernel<CIN, COUT, EPI>), dim3(blocks), dim3(512), lds, st, in,
                       reinterpret_cast<const uint4*>(wpk), wmeta, bias, aux, out, S, tiles, ntiles, flags, prof);
    if (prof) {
        unsigned long long hp[8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hp, prof, 64, hipMemcpyDeviceToHost);
        const double nb = blocks;
        fprintf(stderr, "[v3 prof] per-wave avg memtime ticks: consumer work %.0f wait %.0f | producer work %.0f wait %.0f | weights work %.0f wait %.0f\n",
                hp[0] / (4 * nb), hp[1] / (4 * nb), hp[2] / (3 * nb), hp[3] / (3 * nb), hp[4] / nb, hp[5] / nb);
    }
    return hipGetLastError();
}

hipError_t launch_conv3x3_tile_f16x3_v3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                        const float* bias, const float* aux, float* out, int N, int S, int cin, int cout,
                                        int epi)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
#define T16V3_CASE(CI, CO, EP) \
    if (cin == CI && cout == CO && epi == EP) return launch_tile_f16x3_v3_inst<CI, CO, EP>(st, in, wpk, wmeta, bias, aux, out, N, S);
    T16V3_CASE(64, 64, EPI_BIAS_ELU) T16V3_CASE(64, 64, EPI_MUL_ELUGRAD)
    T16V3_CASE(32, 32, EPI_BIAS_ELU) T16V3_CASE(32, 32, EPI_MUL_ELUGRAD)
#undef T16V3_CASE
    return hipErrorInvalidValue;
}
