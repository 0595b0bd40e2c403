// Retired from the product library in round 3 (was option out_variant=0): the LDS-staged GEMM form of the decoder output conv
// (0.36 ms per launch at cfg3 vs 0.27 for the streaming kernel in kernels_out.hip).  Not compiled.
template <int C>
__global__ __launch_bounds__(256, 2)
void dec_out_gemm_f16x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ wmeta,
                               const float* __restrict__ bias, float4* __restrict__ out, int S, int tiles)
{
    constexpr int NCHUNK = C / 16;
    constexpr int HALO = 18, NPX = HALO * HALO;                     // 324 halo pixels = 11 row-blocks of 32 (last partial)
    constexpr int PXS = 80, IN_BYTES = (NPX + 1) * PXS;
    constexpr int W_U4 = NCHUNK * 2 * 2 * 64;                       // all chunks' weights stay resident (16 KB at C = 64)
    constexpr int NIN = (NPX * 4 + 255) / 256;
    constexpr int PSTR = 36;                                        // floats per pixel in the P tile
    constexpr int P_BYTES = NPX * PSTR * 4;
    static_assert(P_BYTES <= 65536, "P tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* s_in = smem_b;
    uint4* s_w = reinterpret_cast<uint4*>(smem_b + IN_BYTES);
    float* s_max = reinterpret_cast<float*>(smem_b + IN_BYTES + W_U4 * 16);
    float* s_P = reinterpret_cast<float*>(smem_b);                  // aliases the staging area after the last chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, kh = lane >> 5, li = lane & 31;
    int bid = blockIdx.x;
    const int tx = bid % tiles; bid /= tiles;
    const int ty = bid % tiles;
    const int n = bid / tiles;

    int goff[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int idx = tid + k * 256;
        const int px = idx >> 2, cq = idx & 3;
        const int gy = ty * 16 - 1 + px / HALO, gx = tx * 16 - 1 + px % HALO;
        const bool ok = idx < NPX * 4 && gy >= 0 && gy < S && gx >= 0 && gx < S;
        goff[k] = ok ? ((n * S + gy) * S + gx) * C + cq * 4 : -1;
    }
    for (int idx = tid; idx < W_U4; idx += 256) s_w[idx] = wpk[idx];

    // wave w owns row-blocks 3w .. 3w+2 (wave 3: two)
    const int mt0 = 3 * wv, nmt = wv < 3 ? 3 : 2;
    f32x16 acc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float cur_scale = 1.f;
    float4 rin[NIN];
    auto prefetch = [&](int chunk) {
        const float* base = in + chunk * 16;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const bool ok = goff[k] >= 0;
            const float4 v = *reinterpret_cast<const float4*>(base + (ok ? goff[k] : 0));
            rin[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() -> float {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(rin[k].x), fabsf(rin[k].y)), fmaxf(fabsf(rin[k].z), fabsf(rin[k].w))));
        m = wave_max_f32(m);
        if (lane == 0) s_max[wv] = m;
        __syncthreads();
        const float mb = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        const int e = (int)((__float_as_uint(mb) >> 23) & 0xffu) - 127;
        int se = 12 - e;
        se = se > 100 ? 100 : (se < -100 ? -100 : se);
        const float scale = (mb > 0.f && mb < 3.0e38f) ? __uint_as_float((unsigned)(127 + se) << 23) : 1.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int idx = tid + k * 256;
            const int px = idx < NPX * 4 ? idx >> 2 : NPX, cq = idx & 3;
            float4 v = rin[k];
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
        __syncthreads();
        return scale;
    };

    prefetch(0);
    cur_scale = commit();
#pragma unroll
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        if (chunk + 1 < NCHUNK) prefetch(chunk + 1);
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const uint4* q = s_w + ((chunk * 2 + 0) * 2 + kh) * 64 + nt * 32 + li;
            bh[nt] = *reinterpret_cast<const f16x8*>(q);
            bl[nt] = *reinterpret_cast<const f16x8*>(q + 2 * 64);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a < nmt) {
                const int px = (mt0 + a) * 32 + li;
                const unsigned char* p = s_in + (px < NPX ? px : NPX) * PXS + kh * 16;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(p);
                const f16x8 al = *reinterpret_cast<const f16x8*>(p + 32);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[a][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nt], acc[a][nt], 0, 0, 0);
                    acc[a][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nt], acc[a][nt], 0, 0, 0);
                    acc[a][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nt], acc[a][nt], 0, 0, 0);
                }
            }
        }
        if (chunk + 1 < NCHUNK) {
            const float ns = commit();
            if (ns != cur_scale) {
                const float r = ns / cur_scale;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc[a][b][q] *= r;
                cur_scale = ns;
            }
        }
    }
    __syncthreads();                                                // staging area is dead: reuse it for P
    const float inv = wmeta[1] / cur_scale;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (a < nmt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = nt * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = (mt0 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (col < 36 && px < NPX) s_P[px * PSTR + col] = acc[a][nt][r] * inv;
                }
            }
        }
    }
    __syncthreads();
    const int y = tid >> 4, x = tid & 15;
    float4 o = make_float4(bias[0], bias[1], bias[2], bias[3]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float4 v = *reinterpret_cast<const float4*>(s_P + ((y + tap / 3) * HALO + x + tap % 3) * PSTR + tap * 4);
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    }
    out[((size_t)n * S + ty * 16 + y) * S + tx * 16 + x] = o;
}

hipError_t launch_dec_out_gemm_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                     const float* bias, float* out, int N, int S, int C)
{
    IOD_XSKIP(128);
    if (S % 16 != 0) return hipErrorInvalidValue;
    const int tiles = S / 16;
    if (C == 64) {
        constexpr size_t lds = std::max<size_t>((18 * 18 + 1) * 80 + 4 * 2 * 2 * 64 * 16 + 16, 324 * 36 * 4);
        hipLaunchKernelGGL((dec_out_gemm_f16x3_kernel<64>), dim3(N * tiles * tiles), dim3(256), lds, st, in,
                           reinterpret_cast<const uint4*>(wpk), wmeta, bias, reinterpret_cast<float4*>(out), S, tiles);
    } else if (C == 32) {
        constexpr size_t lds = std::max<size_t>((18 * 18 + 1) * 80 + 2 * 2 * 2 * 64 * 16 + 16, 324 * 36 * 4);
        hipLaunchKernelGGL((dec_out_gemm_f16x3_kernel<32>), dim3(N * tiles * tiles), dim3(256), lds, st, in,
                           reinterpret_cast<const uint4*>(wpk), wmeta, bias, reinterpret_cast<float4*>(out), S, tiles);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
