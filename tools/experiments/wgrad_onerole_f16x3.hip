// Retired from the product library in round 3 (was option wgrad_ws=0): the one-role split-fp16 stride-1 weight-gradient kernel.
// Measurements: DESIGN.md 4.2 (1.18-1.22 ms per 64->64 layer vs 0.72-0.75 for the warp-specialised kernel).  Not compiled.

// =========================================================================================
// Split-precision (fp16 hi+lo, 3 MFMAs, fp32 accumulate) stride-1 weight gradient.
//   dW[tap][ci][co] = sum_px a[px+tap][ci] * d[px][co]:  M = ci, N = co, K = pixels (16 per MFMA = one tile row).
// Both operands need 8 K-consecutive (= pixel-consecutive) fp16 values per lane, but NHWC memory is
// channel-contiguous, so the tile is TRANSPOSED while it is staged: LDS holds one fp16 plane per channel
// ([term hi/lo][channel][row][24-half padded row]); the +-1 column shifts of the taps are produced in registers
// (v_alignbit on the aligned 8-pixel vector plus one neighbour element) instead of unaligned LDS reads.
//   tile = 4 x 16 pixels (halo 6 x 18), 4 waves = (ci half, co half[, row split]); persistent blocks.
//   Range: each tile is scaled by powers of two chosen from its max |a|, max |d| (block-local, with hysteresis);
//   the accumulators are rescaled exactly when the product of scales changes.
// =========================================================================================
#ifdef IODINE_TILE_PROF
__device__ unsigned g_wgrad_prof[TP_MAXBLK * 8];
#endif

template <int CI, int NCO>
__global__ __launch_bounds__(256, 2)
void conv3x3_wgrad_f16x3_kernel(const float* __restrict__ a, const float* __restrict__ d, float* __restrict__ part,
                                float* __restrict__ part_b, int S, int ntiles, int tiles_x, int tiles_y)
{
    constexpr int MT = CI / 32;
    constexpr int NTT = (NCO + 31) / 32;
    constexpr int KS = 4 / (MT * NTT);
    constexpr int NCOP = NTT * 32;
    constexpr int TH = 4, HH = TH + 2;
    constexpr int APL = 76, DPL = 36;                    // dwords per channel plane (multiples of 4, odd/4: conflict-free b128)
    constexpr int A4 = CI / 4, D4 = NCO / 4;
    constexpr int NA_UNITS = HH * 10 * A4, ND_UNITS = TH * 8 * D4;
    constexpr int NAU = (NA_UNITS + 255) / 256, NDU = (ND_UNITS + 255) / 256;
    constexpr int RW = TH / KS;                          // tile rows (= K16 steps) per wave

    extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
    unsigned* s_a = smem_u;                              // [2][CI][APL]
    unsigned* s_d = smem_u + 2 * CI * APL;               // [2][NCO][DPL]
    float* s_max = reinterpret_cast<float*>(s_d + 2 * NCO * DPL);     // [8]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    const int mi = wv % MT, ni = (wv / MT) % NTT, ks = wv / (MT * NTT);
    const int ci = mi * 32 + li, co = ni * 32 + li;
    const bool co_ok = NCO >= 32 || li < NCO;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float sa = 1.f, sd = 1.f, acc_prod = 1.f;
    TP_DECL;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const float* a_n = a + (size_t)n * S * S * CI;
        const float* d_n = d + (size_t)n * S * S * NCO;

        // ---- load the tile (pairs of horizontally adjacent pixels, 4 channels each) ----
        float4 ra[NAU][2], rd[NDU][2];
        float ma = 0.f, md = 0.f;
#pragma unroll
        for (int k = 0; k < NAU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % A4, tt = u / A4, p = tt % 10, row = tt / 10;
            const int gy = ty * TH - 1 + row, gx = tx * 16 + 2 * p - 2;
            ra[k][0] = ra[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < NA_UNITS && gy >= 0 && gy < S) {
                if (gx >= 0 && gx < S) ra[k][0] = *reinterpret_cast<const float4*>(a_n + ((size_t)gy * S + gx) * CI + c4 * 4);
                if (gx + 1 >= 0 && gx + 1 < S) ra[k][1] = *reinterpret_cast<const float4*>(a_n + ((size_t)gy * S + gx + 1) * CI + c4 * 4);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ma = fmaxf(ma, fmaxf(fmaxf(fabsf(ra[k][j].x), fabsf(ra[k][j].y)), fmaxf(fabsf(ra[k][j].z), fabsf(ra[k][j].w))));
        }
#pragma unroll
        for (int k = 0; k < NDU; ++k) {
            const int u = tid + k * 256;
            const int c4 = u % D4, tt = u / D4, p = tt % 8, row = tt / 8;
            const int gy = ty * TH + row, gx = tx * 16 + 2 * p;
            rd[k][0] = rd[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < ND_UNITS) {
                rd[k][0] = *reinterpret_cast<const float4*>(d_n + ((size_t)gy * S + gx) * NCO + c4 * 4);
                rd[k][1] = *reinterpret_cast<const float4*>(d_n + ((size_t)gy * S + gx + 1) * NCO + c4 * 4);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                md = fmaxf(md, fmaxf(fmaxf(fabsf(rd[k][j].x), fabsf(rd[k][j].y)), fmaxf(fabsf(rd[k][j].z), fabsf(rd[k][j].w))));
                bsum.x += rd[k][j].x; bsum.y += rd[k][j].y; bsum.z += rd[k][j].z; bsum.w += rd[k][j].w;
            }
        }
        ma = wave_max_f32(ma);
        md = wave_max_f32(md);
        TP_STAMP(0);                                       // [0] tile loads issued, arrived, max
        if (lane == 0) { s_max[wv] = ma; s_max[4 + wv] = md; }
        __syncthreads();                                   // every wave is also done with the previous tile's planes
        TP_STAMP(1);                                       // [1] barrier 1
        ma = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        md = fmaxf(fmaxf(s_max[4], s_max[5]), fmaxf(s_max[6], s_max[7]));
        sa = tile_scale(ma, sa);
        sd = tile_scale(md, sd);
        const float prod = sa * sd;
        if (prod != acc_prod) {                            // block-uniform; exact (powers of two)
            const float r = prod / acc_prod;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[tp][q] *= r;
            acc_prod = prod;
        }

        // ---- transpose into per-channel fp16 planes (dword = 2 horizontally adjacent pixels) ----
#pragma unroll
        for (int k = 0; k < NAU; ++k) {
            const int u = tid + k * 256;
            if (u < NA_UNITS) {
                const int c4 = u % A4, tt = u / A4, p = tt % 10, row = tt / 10;
                const int rot = c4 & 3;                    // rotate so that 16 lanes hit 8 banks (2-way = free)
                const float4 q0 = rot4(ra[k][0], rot), q1 = rot4(ra[k][1], rot);
                const float x0[4] = {q0.x, q0.y, q0.z, q0.w}, x1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned lo;
                    const unsigned hi = pack_hi_lo(x0[e] * sa, x1[e] * sa, lo);
                    const int ch = c4 * 4 + ((e + rot) & 3);
                    s_a[(0 * CI + ch) * APL + row * 12 + p + 3] = hi;
                    s_a[(1 * CI + ch) * APL + row * 12 + p + 3] = lo;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NDU; ++k) {
            const int u = tid + k * 256;
            if (u < ND_UNITS) {
                const int c4 = u % D4, tt = u / D4, p = tt % 8, row = tt / 8;
                const int rot = c4 & 3;
                const float4 q0 = rot4(rd[k][0], rot), q1 = rot4(rd[k][1], rot);
                const float x0[4] = {q0.x, q0.y, q0.z, q0.w}, x1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned lo;
                    const unsigned hi = pack_hi_lo(x0[e] * sd, x1[e] * sd, lo);
                    const int ch = c4 * 4 + ((e + rot) & 3);
                    s_d[(0 * NCO + ch) * DPL + row * 8 + p] = hi;
                    s_d[(1 * NCO + ch) * DPL + row * 8 + p] = lo;
                }
            }
        }
        TP_STAMP(2);                                       // [2] scale, split, transposed LDS writes
        __syncthreads();
        TP_STAMP(3);                                       // [3] barrier 2

        // ---- MFMA: one K=16 step per tile row; taps = 3 halo rows x 3 register-shifted column variants ----
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = ks * RW + rr;
            h16x8 bh, bl;
            {
                uint4 vb_h = make_uint4(0, 0, 0, 0), vb_l = make_uint4(0, 0, 0, 0);
                if (co_ok) {
                    vb_h = *reinterpret_cast<const uint4*>(s_d + (0 * NCO + co) * DPL + r * 8 + 4 * kh);
                    vb_l = *reinterpret_cast<const uint4*>(s_d + (1 * NCO + co) * DPL + r * 8 + 4 * kh);
                }
                __builtin_memcpy(&bh, &vb_h, 16); __builtin_memcpy(&bl, &vb_l, 16);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                h16x8 A[2][3];                               // [term][dx]
#pragma unroll
                for (int term = 0; term < 2; ++term) {
                    const unsigned* pl = s_a + (term * CI + ci) * APL + (r + dy) * 12;
                    const uint4 v = *reinterpret_cast<const uint4*>(pl + 4 + 4 * kh);          // columns 8kh .. 8kh+7
                    const unsigned prev = pl[3 + 4 * kh] & 0xffff0000u;                        // column 8kh-1 in the high half
                    const unsigned next = pl[8 + 4 * kh] & 0x0000ffffu;                        // column 8kh+8 in the low half
                    uint4 m1, p1;
                    m1.x = __builtin_amdgcn_alignbit(v.x, prev, 16);                           // [c-1, c0]
                    m1.y = __builtin_amdgcn_alignbit(v.y, v.x, 16);
                    m1.z = __builtin_amdgcn_alignbit(v.z, v.y, 16);
                    m1.w = __builtin_amdgcn_alignbit(v.w, v.z, 16);
                    p1.x = __builtin_amdgcn_alignbit(v.y, v.x, 16);                            // [c1, c2]
                    p1.y = __builtin_amdgcn_alignbit(v.z, v.y, 16);
                    p1.z = __builtin_amdgcn_alignbit(v.w, v.z, 16);
                    p1.w = __builtin_amdgcn_alignbit(next, v.w, 16);                           // [c7, c8]
                    __builtin_memcpy(&A[term][0], &m1, 16);
                    __builtin_memcpy(&A[term][1], &v, 16);
                    __builtin_memcpy(&A[term][2], &p1, 16);
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tap = dy * 3 + dx;
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][dx], bh, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][dx], bl, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][dx], bh, acc[tap], 0, 0, 0);
                }
            }
        }
        TP_STAMP(4);                                       // [4] LDS fragment reads + shifts + 27 MFMAs per tile row
    }
    TP_FLUSH(g_wgrad_prof);

    const float inv = 1.f / acc_prod;
    float* pw = part + ((size_t)(blockIdx.x * KS + ks) * 9) * CI * NCOP;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            pw[((size_t)tap * CI + cr) * NCOP + ni * 32 + li] = acc[tap][r] * inv;
        }
    __syncthreads();
    float4* s_red = reinterpret_cast<float4*>(smem_u);
    s_red[tid] = bsum;
    __syncthreads();
    if (tid < D4) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = tid; j < 256; j += D4) { const float4 v = s_red[j]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
        *reinterpret_cast<float4*>(part_b + (size_t)blockIdx.x * NCO + tid * 4) = t4;
    }
}

int wgrad_f16_blocks(int N, int S) { const int nt = N * (S / 4) * (S / 16); return nt < 512 ? nt : 512; }

template <int CI, int NCO>
static hipError_t launch_wgrad_f16_inst(hipStream_t st, const float* a, const float* d, float* part, float* part_b,
                                        int N, int S, int* nparts, int* ncop)
{
    constexpr int MT = CI / 32, NTT = (NCO + 31) / 32, KS = 4 / (MT * NTT);
    constexpr size_t lds = (size_t)(2 * CI * 76 + 2 * NCO * 36) * 4 + 32;
    static std::atomic<unsigned> attr_devs{0};                             // devices this instance is configured on
    if (hipError_t e = iod_set_max_lds((const void*)conv3x3_wgrad_f16x3_kernel<CI, NCO>, (int)lds, attr_devs); e != hipSuccess) return e;
    const int tiles_x = S / 16, tiles_y = S / 4, ntiles = N * tiles_x * tiles_y;
    const int blocks = wgrad_f16_blocks(N, S);
    hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<CI, NCO>), dim3(blocks), dim3(256), lds, st, a, d, part, part_b, S,
                       ntiles, tiles_x, tiles_y);
#ifdef IODINE_TILE_PROF
    {
        std::vector<unsigned> hp((size_t)blocks * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_wgrad_prof), hp.size() * sizeof(unsigned));
        static const char* names[5] = {"loads+max", "barrier1", "split+transposed-lds-write", "barrier2", "taps(lds-read+shift+mfma)"};
        double sum[5] = {0}, tot = 0;
        for (int b2 = 0; b2 < blocks; ++b2) for (int i = 0; i < 5; ++i) sum[i] += hp[(size_t)b2 * 8 + i];
        const double per = (double)blocks * ((double)ntiles / blocks);
        for (int i = 0; i < 5; ++i) tot += sum[i] / per;
        fprintf(stderr, "[wgrad prof <%d,%d>] memtime ticks per TILE (thread 0), total %.0f:", CI, NCO, tot);
        for (int i = 0; i < 5; ++i) fprintf(stderr, " %s %.0f |", names[i], sum[i] / per);
        fprintf(stderr, "\n");
    }
#endif
    *nparts = blocks * KS;
    *ncop = NTT * 32;
    return hipGetLastError();
}

hipError_t launch_conv3x3_wgrad_f16x3(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N,
                                      int S, int ci, int nco, int* nparts, int* ncop, int* nbias_parts)
{
    if (S % 16 != 0) return hipErrorInvalidValue;
    *nbias_parts = wgrad_f16_blocks(N, S);
    if (ci == 64 && nco == 64) return launch_wgrad_f16_inst<64, 64>(st, a, d, part, part_b, N, S, nparts, ncop);
    if (ci == 32 && nco == 32) return launch_wgrad_f16_inst<32, 32>(st, a, d, part, part_b, N, S, nparts, ncop);
    if (ci == 64 && nco == 4) return launch_wgrad_f16_inst<64, 4>(st, a, d, part, part_b, N, S, nparts, ncop);
    if (ci == 32 && nco == 4) return launch_wgrad_f16_inst<32, 4>(st, a, d, part, part_b, N, S, nparts, ncop);
    return hipErrorInvalidValue;
}
