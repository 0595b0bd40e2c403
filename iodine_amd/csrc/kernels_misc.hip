// Small kernels around the conv stacks: spatial-broadcast first decoder layer and its backward
// reductions, posterior sampling / KL / ELBO, latent layer-norm, refinement head (avg-pool, MLP,
// LSTM cell, posterior update).  All are HBM- or latency-bound; none is GEMM-shaped enough for MFMA
// at the slot-batch sizes involved (N = B*K rows).
#include "common.h"

// -----------------------------------------------------------------------------------------------
// layout conversion: images NCHW (B,3,P) -> (B,P) float4 {r,g,b,0}
// -----------------------------------------------------------------------------------------------
__global__ void x_to_nhwc4_kernel(const float* __restrict__ x, float4* __restrict__ x4, int B, int P)
{
    const size_t total = (size_t)B * P;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / P, p = i % P;
        const float* xb = x + b * 3 * P;
        x4[i] = make_float4(xb[p], xb[P + p], xb[2 * (size_t)P + p], 0.f);
    }
}

hipError_t launch_x_to_nhwc4(hipStream_t st, const float* x, float* x4, int B, int P)
{
    const int blocks = (int)std::min<size_t>(((size_t)B * P + 255) / 256, 4096);
    hipLaunchKernelGGL(x_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, st, x, (float4*)x4, B, P);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// Gaussian.init_unit (lib/modeling/iodine.py:607-618): lambda = learned init, LSTM state = 0
// -----------------------------------------------------------------------------------------------
__global__ void posterior_init_kernel(const float* __restrict__ init_mean, const float* __restrict__ init_logvar,
                                      float* __restrict__ pm, float* __restrict__ plv, float* __restrict__ h,
                                      float* __restrict__ c, int N, int L, int H)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * L) { pm[i] = init_mean[i % L]; plv[i] = init_logvar[i % L]; }
    if (i < N * H) { h[i] = 0.f; c[i] = 0.f; }
}

hipError_t launch_posterior_init(hipStream_t st, const float* im, const float* ilv, float* pm, float* plv, float* h,
                                 float* c, int N, int L, int H)
{
    const int tot = N * (L > H ? L : H);
    hipLaunchKernelGGL(posterior_init_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, im, ilv, pm, plv, h, c, N, L, H);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// Decoder layer 0 (SpatialBroadcast + first conv, iodine.py:512-540,583,592) without materialising
// the (N, L+2, S, S) broadcast: z is spatially constant, so
//     a0[n,p,co] = ELU( V[n, cls(p), co] + Cmap[p, co] ),  V = z . Wcls  (9 border classes),
//     Wcls[cls][ci][co] = sum over the taps that stay inside the image for that class of W[co][ci][tap]
//     Cmap[p][co] = bias[co] + conv(coordinate planes)[p][co]   (depends on the weights only).
// -----------------------------------------------------------------------------------------------
IOD_DEVINL bool tap_valid_for_class(int cls3, int d) { return !((cls3 == 0 && d == 0) || (cls3 == 2 && d == 2)); }

__global__ void dec_l0_prepare_kernel(const float* __restrict__ w /*[C][L+2][3][3]*/, const float* __restrict__ bias,
                                      const float* __restrict__ lin, int C, int L, int S,
                                      float* __restrict__ wcls /*[9][L][C]*/, float* __restrict__ wclsT /*[9][C][L]*/,
                                      float* __restrict__ cmap /*[P][C]*/)
{
    const int I = L + 2;
    const size_t n_w = (size_t)9 * L * C, n_c = (size_t)S * S * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_w + n_c;
         idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < n_w) {
            const int co = idx % C, ci = (idx / C) % L, cls = idx / ((size_t)C * L);
            const int rc = cls / 3, cc = cls % 3;
            float s = 0.f;
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx)
                    if (tap_valid_for_class(rc, dy) && tap_valid_for_class(cc, dx))
                        s += w[((size_t)co * I + ci) * 9 + dy * 3 + dx];
            wcls[idx] = s;
            wclsT[((size_t)cls * C + co) * L + ci] = s;
        } else {
            const size_t j = idx - n_w;
            const int co = j % C, p = j / C;
            const int y = p / S, x = p % S;
            float s = bias[co];
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    const int yy = y + dy - 1, xx = x + dx - 1;
                    if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
                    s += lin[xx] * w[((size_t)co * I + L) * 9 + dy * 3 + dx];        // channel L   = x coordinate
                    s += lin[yy] * w[((size_t)co * I + L + 1) * 9 + dy * 3 + dx];    // channel L+1 = y coordinate
                }
            cmap[j] = s;
        }
    }
}

hipError_t launch_dec_l0_prepare(hipStream_t st, const float* w, const float* bias, const float* lin, int C, int L,
                                 int S, float* wcls, float* wclsT, float* cmap)
{
    hipLaunchKernelGGL(dec_l0_prepare_kernel, dim3(1024), dim3(256), 0, st, w, bias, lin, C, L, S, wcls, wclsT, cmap);
    return hipGetLastError();
}

// z = mean + exp(logvar/2) * eps (Gaussian.sample, iodine.py:620-634), then V = z . Wcls.  One block per slot.
__global__ __launch_bounds__(1024)
void dec_v_kernel(const float* __restrict__ pm, const float* __restrict__ plv, const float* __restrict__ eps,
                  const float* __restrict__ z_in, const float* __restrict__ wcls, float* __restrict__ z_out,
                  float* __restrict__ V, int L, int C)
{
    extern __shared__ float s_z[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int nth = blockDim.x;
    for (int l = tid; l < L; l += nth) {
        float z;
        if (z_in) z = z_in[(size_t)n * L + l];
        else z = pm[(size_t)n * L + l] + expf(0.5f * plv[(size_t)n * L + l]) * eps[(size_t)n * L + l];
        s_z[l] = z;
        if (z_out) z_out[(size_t)n * L + l] = z;
    }
    __syncthreads();
    for (int o = tid; o < 9 * C; o += nth) {             // (one output per thread at C = 64: the loop is a chain of L2 load latencies)
        const int cls = o / C, co = o % C;
        const float* w = wcls + (size_t)cls * L * C + co;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // eight partial sums: the loads of a chain overlap
        int ci = 0;
        for (; ci + 7 < L; ci += 8) {
            float wv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) wv[q] = w[(size_t)(ci + q) * C];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = fmaf(s_z[ci + q], wv[q], a[q]);
        }
        for (; ci < L; ++ci) a[0] = fmaf(s_z[ci], w[(size_t)ci * C], a[0]);
        V[(size_t)n * 9 * C + o] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
}

hipError_t launch_dec_v(hipStream_t st, const float* pm, const float* plv, const float* eps, const float* z_in,
                        const float* wcls, float* z_out, float* V, int N, int L, int C)
{
    const int nth = std::min(1024, (9 * C + 63) / 64 * 64);
    hipLaunchKernelGGL(dec_v_kernel, dim3(N), dim3(nth), L * sizeof(float), st, pm, plv, eps, z_in, wcls, z_out, V, L, C);
    return hipGetLastError();
}

// act0[n][p][:] = ELU(V[n][class(p)][:] + cmap[p][:]).  Pure streaming write (N*P*C floats): grid.y = slot, 32-bit index
// arithmetic only, four float4 stores in flight per thread.
__global__ __launch_bounds__(256)
void dec_l0_kernel(const float4* __restrict__ V, const float4* __restrict__ cmap, float4* __restrict__ out,
                   int S, int C4, int pc4)
{
    const int n = blockIdx.y;
    const float4* Vn = V + (size_t)n * 9 * C4;
    float4* on = out + (size_t)n * pc4;
    const int base = blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + k * 256;
        if (i < pc4) {
            const int p = i / C4, c4 = i - p * C4;
            const int y = p / S, x = p - y * S;
            const int cls = (y == 0 ? 0 : (y == S - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == S - 1 ? 2 : 1));
            const float4 v = Vn[cls * C4 + c4];
            const float4 m = cmap[i];
            on[i] = make_float4(elu1_fast(v.x + m.x), elu1_fast(v.y + m.y), elu1_fast(v.z + m.z), elu1_fast(v.w + m.w));
        }
    }
}

// The same with one block per 8 x 16 CELL of a slot-image (16 pixels x C floats = one contiguous run per row), which also
// leaves the cell's max |act0| in the side buffer the weight-stationary conv reads its tile scale from (tmax[n][cell][4]).
template <int C4>
__global__ __launch_bounds__(256)
void dec_l0_cells_kernel(const float4* __restrict__ V, const float4* __restrict__ cmap, float4* __restrict__ out,
                         float* __restrict__ tmax, int S)
{
    const int cells_x = S >> 4;
    const int cx = blockIdx.x % cells_x, cy = blockIdx.x / cells_x, n = blockIdx.y;
    const float4* Vn = V + (size_t)n * 9 * C4;
    float4* on = out + (size_t)n * S * S * C4;
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < 128 * C4 / 256; ++k) {
        const int i = threadIdx.x + k * 256;
        const int px = i / C4, c4 = i % C4;
        const int y = cy * 8 + (px >> 4), x = cx * 16 + (px & 15);
        const int cls = (y == 0 ? 0 : (y == S - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == S - 1 ? 2 : 1));
        const int gi = (y * S + x) * C4 + c4;
        const float4 v = Vn[cls * C4 + c4];
        const float4 m = cmap[gi];
        const float4 o = make_float4(elu1_fast(v.x + m.x), elu1_fast(v.y + m.y), elu1_fast(v.z + m.z), elu1_fast(v.w + m.w));
        on[gi] = o;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
    }
    mx = wave_max_f32(mx);
    if ((threadIdx.x & 63) == 0) tmax[((size_t)n * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = mx;
}

hipError_t launch_dec_l0(hipStream_t st, const float* V, const float* cmap, float* out, int N, int S, int C, float* tmax)
{
    IOD_XSKIP(16);
    if (tmax && S % 16 == 0 && (C == 64 || C == 32)) {
        const dim3 grid((S / 16) * (S / 8), N);
        if (C == 64) hipLaunchKernelGGL(dec_l0_cells_kernel<16>, grid, dim3(256), 0, st, (const float4*)V, (const float4*)cmap, (float4*)out, tmax, S);
        else hipLaunchKernelGGL(dec_l0_cells_kernel<8>, grid, dim3(256), 0, st, (const float4*)V, (const float4*)cmap, (float4*)out, tmax, S);
        return hipGetLastError();
    }
    const int pc4 = S * S * (C / 4);
    hipLaunchKernelGGL(dec_l0_kernel, dim3((pc4 + 1023) / 1024, N), dim3(256), 0, st, (const float4*)V,
                       (const float4*)cmap, (float4*)out, S, C / 4, pc4);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// Backward of decoder layer 0 wrt z: class sums of dpre0 over pixels.
//   stage 1: one block per (n, row): left / interior / right column sums   -> rows[n][y][3][C]
//   stage 2: one block per n: top / middle / bottom over rows              -> Rc[n][9][C]
// -----------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256)
void l0_reduce_rows_kernel(const float4* __restrict__ dpre, float4* __restrict__ rows, int S)
{
    constexpr int C4 = C / 4, PL = 256 / C4;            // float4 lanes per pixel, pixel lanes
    const int y = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int c4 = tid % C4, pl = tid / C4;
    const float4* src = dpre + ((size_t)n * S + y) * S * C4;
    float4 mid = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int x = 1 + pl; x < S - 1; x += PL) {
        const float4 v = src[(size_t)x * C4 + c4];
        mid.x += v.x; mid.y += v.y; mid.z += v.z; mid.w += v.w;
    }
    __shared__ float4 s_red[256];
    s_red[tid] = mid;
    __syncthreads();
    for (int s = PL / 2; s > 0; s >>= 1) {
        if (pl < s) {
            float4 a = s_red[tid], b = s_red[tid + s * C4];
            s_red[tid] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        __syncthreads();
    }
    if (tid < C4) {
        float4* o = rows + ((size_t)n * S + y) * 3 * C4;
        o[0 * C4 + tid] = src[tid];                              // x = 0
        o[1 * C4 + tid] = s_red[tid];                            // 1 .. S-2
        o[2 * C4 + tid] = src[(size_t)(S - 1) * C4 + tid];       // x = S-1
    }
}

// blockDim = 4 * 3C: four row slices per (cc, co) column, four independent partial sums each, combined in fixed order
// (the one-thread-per-column form walked the S - 2 interior rows as one dependent chain: 52 us for 22 MB at cfg3)
__global__ void l0_reduce_cls_kernel(const float* __restrict__ rows, float* __restrict__ Rc, int S, int C)
{
    __shared__ float s_mid[4][192];
    const int n = blockIdx.x, W = 3 * C;
    const int t = threadIdx.x % W, slice = threadIdx.x / W;      // t = cc*C + co
    const float* src = rows + (size_t)n * S * W + t;
    const int per = (S - 2 + 3) / 4, y0 = 1 + slice * per, y1 = min(S - 1, y0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int y = y0;
    for (; y + 3 < y1; y += 4) {
        a0 += src[(size_t)y * W]; a1 += src[(size_t)(y + 1) * W]; a2 += src[(size_t)(y + 2) * W]; a3 += src[(size_t)(y + 3) * W];
    }
    for (; y < y1; ++y) a0 += src[(size_t)y * W];
    s_mid[slice][t] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (slice == 0) {
        float* o = Rc + (size_t)n * 9 * C;
        const int cc = t / C, co = t % C;
        o[(0 * 3 + cc) * C + co] = src[0];
        o[(1 * 3 + cc) * C + co] = (s_mid[0][t] + s_mid[1][t]) + (s_mid[2][t] + s_mid[3][t]);
        o[(2 * 3 + cc) * C + co] = src[(size_t)(S - 1) * W];
    }
}

// The same from per-tile row sums rows_p[n][y][tile x][NQ][C] (written by the EPI_L0ROWS / EPI_L0ROWSX epilogue of the layer-1
// data gradient, which then never stores d(pre-activation 0)): sum over the tile columns first, then over the rows by class.
//   NQ = 3 (inference): left / interior / right column sums.
//   NQ = 4 (training): + the x-coordinate-weighted sum; the per-row sums over the tile columns also go to rown[n][y][4][C] -
//     their sum over the slot-images (l0_rowsum_acc_kernel) is all the coordinate-channel and bias gradients of the broadcast
//     layer need, so d(pre-activation 0) is never stored in training either.
// Grid (slot-image, row slice): one block per slot-image left 224 blocks of the 256 CUs reading 176 - 235 MB at 2.4 TB/s
// (72 - 83 us); with L0R_SLICES row slices per slot-image and float4 columns the interior-row sums come as partials that a second, tiny
// kernel adds in fixed order.
constexpr int L0R_SLICES = 16;
template <int NQ>
__global__ __launch_bounds__(256)
void l0_rows_reduce_kernel(const float4* __restrict__ rows_p, float4* __restrict__ part, float4* __restrict__ edge,
                           float4* __restrict__ rown, int S, int C, int tiles)
{
    __shared__ float4 s_mid[4][64];
    const int n = blockIdx.x, sl = blockIdx.y, W4 = NQ * C / 4, C4 = C / 4;
    const int t = threadIdx.x % W4, sub = threadIdx.x / W4;      // t = float4 column: q*C/4 + co/4; four row sub-slices
    const float4* src = rows_p + (size_t)n * S * tiles * W4 + t;
    float4* dst = rown ? rown + (size_t)n * S * W4 + t : nullptr;
    auto add4 = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    auto row_sum = [&](int y) {
        const float4* p = src + (size_t)y * tiles * W4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        int tx = 0;
        for (; tx + 1 < tiles; tx += 2) { add4(a, p[(size_t)tx * W4]); add4(b, p[(size_t)(tx + 1) * W4]); }
        if (tx < tiles) add4(a, p[(size_t)tx * W4]);
        add4(a, b);
        return a;
    };
    const int per = (S - 2 + L0R_SLICES - 1) / L0R_SLICES;
    const int y0 = 1 + sl * per, y1 = min(S - 1, y0 + per);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    int y = y0 + sub;
    for (; y + 4 < y1; y += 8) {
        const float4 r0 = row_sum(y), r1 = row_sum(y + 4);
        if (dst) { dst[(size_t)y * W4] = r0; dst[(size_t)(y + 4) * W4] = r1; }
        add4(a0, r0); add4(a1, r1);
    }
    if (y < y1) { const float4 r0 = row_sum(y); if (dst) dst[(size_t)y * W4] = r0; add4(a0, r0); }
    add4(a0, a1);
    s_mid[sub][t] = a0;
    __syncthreads();
    if (sub == 0 && t < 3 * C4) {
        float4 m0 = s_mid[0][t], m2 = s_mid[2][t];
        add4(m0, s_mid[1][t]); add4(m2, s_mid[3][t]); add4(m0, m2);
        part[((size_t)n * L0R_SLICES + sl) * 3 * C4 + t] = m0;
    }
    // first / last image row: their own classes
    if (sl == 0 && sub == 1) {
        const float4 top = row_sum(0);
        if (dst) dst[0] = top;
        if (t < 3 * C4) edge[((size_t)n * 2 + 0) * 3 * C4 + t] = top;
    }
    if (sl == L0R_SLICES - 1 && sub == 2) {
        const float4 bot = row_sum(S - 1);
        if (dst) dst[(size_t)(S - 1) * W4] = bot;
        if (t < 3 * C4) edge[((size_t)n * 2 + 1) * 3 * C4 + t] = bot;
    }
}

// Rc[n][row class][column class][C]: top row, interior rows (the slices' partials in fixed order), bottom row
__global__ void l0_rows_combine_kernel(const float* __restrict__ part, const float* __restrict__ edge, float* __restrict__ Rc, int C)
{
    const int n = blockIdx.x, t = threadIdx.x;                   // t = cc*C + co
    if (t >= 3 * C) return;
    const float* p = part + (size_t)n * L0R_SLICES * 3 * C + t;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int s = 0; s < L0R_SLICES; s += 2) { a += p[(size_t)s * 3 * C]; b += p[(size_t)(s + 1) * 3 * C]; }
    float* o = Rc + (size_t)n * 9 * C;
    o[0 * 3 * C + t] = edge[((size_t)n * 2 + 0) * 3 * C + t];
    o[1 * 3 * C + t] = a + b;
    o[2 * 3 * C + t] = edge[((size_t)n * 2 + 1) * 3 * C + t];
}

// Rsum[i] = (first ? 0 : Rsum[i]) + alpha * sum_n rown[n][i], i < len: fixed order over the slot-images (sixteen interleaved partial
// sums), one thread per element
__global__ void l0_rowsum_acc_kernel(const float* __restrict__ rown, int N, int len, float alpha, int first, float* __restrict__ Rsum)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const float* p = rown + i;
    float a[16];                                             // sixteen loads in flight: the loop is a chain of memory latencies
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = 0.f;
    int n = 0;
    for (; n + 15 < N; n += 16) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = p[(size_t)(n + q) * len];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] += v[q];
    }
    for (; n < N; ++n) a[n & 15] += p[(size_t)n * len];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int q = 0; q < w; ++q) a[q] += a[q + w];
    const float s = alpha * a[0];
    Rsum[i] = first ? s : Rsum[i] + s;
}

// scratch: l0_rows_scratch_floats(N, C) floats
size_t l0_rows_scratch_floats(int N, int C) { return (size_t)N * (L0R_SLICES + 2) * 3 * C; }

hipError_t launch_l0_reduce_cls_tiles(hipStream_t st, const float* rows_p, float* Rc, int N, int S, int C, float* scratch)
{
    IOD_XSKIP(32);
    if (C > 64 || S % 16 != 0 || S < 4 || !scratch) return hipErrorInvalidValue;
    float *part = scratch, *edge = scratch + (size_t)N * L0R_SLICES * 3 * C;
    hipLaunchKernelGGL(l0_rows_reduce_kernel<3>, dim3(N, L0R_SLICES), dim3(3 * C), 0, st, (const float4*)rows_p, (float4*)part, (float4*)edge,
                       (float4*)nullptr, S, C, S / 16);
    hipLaunchKernelGGL(l0_rows_combine_kernel, dim3(N), dim3(3 * C), 0, st, part, edge, Rc, C);
    return hipGetLastError();
}

hipError_t launch_l0_reduce_cls_tiles_x(hipStream_t st, const float* rows_p, float* Rc, float* rown, int N, int S, int C, float* scratch)
{
    IOD_XSKIP(32);
    if (C > 64 || S % 16 != 0 || S < 4 || !scratch) return hipErrorInvalidValue;
    float *part = scratch, *edge = scratch + (size_t)N * L0R_SLICES * 3 * C;
    hipLaunchKernelGGL(l0_rows_reduce_kernel<4>, dim3(N, L0R_SLICES), dim3(4 * C), 0, st, (const float4*)rows_p, (float4*)part, (float4*)edge,
                       (float4*)rown, S, C, S / 16);
    hipLaunchKernelGGL(l0_rows_combine_kernel, dim3(N), dim3(3 * C), 0, st, part, edge, Rc, C);
    return hipGetLastError();
}

hipError_t launch_l0_rowsum_acc(hipStream_t st, const float* rown, int N, int S, int C, float alpha, int first, float* Rsum)
{
    const int len = S * 4 * C;
    hipLaunchKernelGGL(l0_rowsum_acc_kernel, dim3((len + 255) / 256), dim3(256), 0, st, rown, N, len, alpha, first, Rsum);
    return hipGetLastError();
}

// Fused variant for rows of exactly NIT*256 float4: one block per (image row y, slot group g) walks the slots of its
// group, reading dpre0 ONCE for both consumers:
//   rows[n][y][3][C]   left / interior / right sums of this row (as above)
//   Dpart[g][p][C]     sum over the group's slots (training only: layer-0 coordinate/bias gradients need sum_n dpre0)
// All NIT loads of a slot are issued before they are consumed; the cross-pixel-lane reduction is shuffles inside a
// wave plus one LDS pass per block (not per slot).
template <int C, int NIT, bool WITH_D>
__global__ __launch_bounds__(256)
void l0_reduce_fused_kernel(const float4* __restrict__ dpre, float4* __restrict__ rows, float4* __restrict__ Dpart,
                            int S, int N, int per_group)
{
    constexpr int C4 = C / 4, PL = 256 / C4, WPL = 64 / C4;      // pixel lanes per block / per wave
    constexpr int MAXG = 32;
    __shared__ float4 s_part[MAXG][4][C4];
    const int y = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int c4 = tid % C4, pl = tid / C4, wv = tid >> 6;
    const int n0 = g * per_group, n1 = min(N, n0 + per_group);
    float4 dacc[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) dacc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = n0; n < n1; ++n) {
        const float4* src = dpre + ((size_t)n * S + y) * S * C4;
        float4 v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) v[k] = src[tid + k * 256];
        float4 mid = make_float4(0.f, 0.f, 0.f, 0.f);
        float4* o = rows + ((size_t)n * S + y) * 3 * C4;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (WITH_D) { dacc[k].x += v[k].x; dacc[k].y += v[k].y; dacc[k].z += v[k].z; dacc[k].w += v[k].w; }
            const int x = pl + k * PL;
            if (x == 0) o[c4] = v[k];
            else if (x == S - 1) o[2 * C4 + c4] = v[k];
            else { mid.x += v[k].x; mid.y += v[k].y; mid.z += v[k].z; mid.w += v[k].w; }
        }
#pragma unroll
        for (int off = C4; off < 64; off <<= 1) {
            mid.x += __shfl_xor(mid.x, off, 64); mid.y += __shfl_xor(mid.y, off, 64);
            mid.z += __shfl_xor(mid.z, off, 64); mid.w += __shfl_xor(mid.w, off, 64);
        }
        if ((tid & 63) < C4) s_part[n - n0][wv][c4] = mid;
    }
    __syncthreads();
    for (int j = tid; j < (n1 - n0) * C4; j += 256) {
        const int nl = j / C4, cc = j % C4;
        const float4 a = s_part[nl][0][cc], b2 = s_part[nl][1][cc], c2 = s_part[nl][2][cc], d2 = s_part[nl][3][cc];
        rows[((size_t)(n0 + nl) * S + y) * 3 * C4 + C4 + cc] =
            make_float4((a.x + b2.x) + (c2.x + d2.x), (a.y + b2.y) + (c2.y + d2.y), (a.z + b2.z) + (c2.z + d2.z),
                        (a.w + b2.w) + (c2.w + d2.w));
    }
    if (WITH_D) {
        float4* dst = Dpart + ((size_t)g * S + y) * S * C4;
#pragma unroll
        for (int k = 0; k < NIT; ++k) dst[tid + k * 256] = dacc[k];
    }
    (void)WPL;
}

// Dacc = (first ? 0 : Dacc) + alpha * sum_g Dpart[g]
__global__ void d_accumulate_kernel(const float4* __restrict__ Dpart, int G, int pc4, float alpha, int first,
                                    float4* __restrict__ Dacc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pc4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < G; ++g) {
        const float4 v = Dpart[(size_t)g * pc4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4 o = first ? make_float4(0.f, 0.f, 0.f, 0.f) : Dacc[i];
    o.x = fmaf(alpha, s.x, o.x); o.y = fmaf(alpha, s.y, o.y); o.z = fmaf(alpha, s.z, o.z); o.w = fmaf(alpha, s.w, o.w);
    Dacc[i] = o;
}

// D[p][c] = sum_n dpre0[n][p][c]  (generic fallback of the fused kernel's Dpart)
__global__ void sum_over_slots_kernel(const float4* __restrict__ dpre, float4* __restrict__ D, int N, size_t pc4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pc4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = 0; n < N; ++n) {
        const float4 v = dpre[(size_t)n * pc4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    D[i] = s;
}

// Row / class sums of dpre0 (always) and, when Dpart != nullptr, Dacc (+)= alpha * sum_n dpre0 (training).
// Dpart must hold L0_DGROUPS * P * C floats.
hipError_t launch_l0_reduce(hipStream_t st, const float* dpre, float* rows, float* Rc, int N, int S, int C, float* Dpart,
                            float* Dacc, float alpha, int first)
{
    IOD_XSKIP(32);
    const int pc4 = S * S * (C / 4);
    const int per_group = (N + l0_dgroups(N) - 1) / l0_dgroups(N);   // <= 32
    const int Gr = (N + per_group - 1) / per_group;              // groups actually launched
    bool fused = false;
#define L0F_CASE(CC, SS)                                                                                             \
    if (C == CC && S == SS) {                                                                                       \
        constexpr int NIT = SS * (CC / 4) / 256;                                                                     \
        if (Dpart)                                                                                                   \
            hipLaunchKernelGGL((l0_reduce_fused_kernel<CC, NIT, true>), dim3(S, Gr), dim3(256), 0, st,              \
                               (const float4*)dpre, (float4*)rows, (float4*)Dpart, S, N, per_group);                \
        else                                                                                                         \
            hipLaunchKernelGGL((l0_reduce_fused_kernel<CC, NIT, false>), dim3(S, Gr), dim3(256), 0, st,             \
                               (const float4*)dpre, (float4*)rows, (float4*)nullptr, S, N, per_group);              \
        fused = true;                                                                                                \
    }
    L0F_CASE(64, 128) else L0F_CASE(32, 64) else L0F_CASE(64, 64) else L0F_CASE(32, 128)
#undef L0F_CASE
    int Gd = Gr;
    if (!fused) {
        if (C == 64)
            hipLaunchKernelGGL((l0_reduce_rows_kernel<64>), dim3(S, N), dim3(256), 0, st, (const float4*)dpre, (float4*)rows, S);
        else if (C == 32)
            hipLaunchKernelGGL((l0_reduce_rows_kernel<32>), dim3(S, N), dim3(256), 0, st, (const float4*)dpre, (float4*)rows, S);
        else
            return hipErrorInvalidValue;
        if (Dpart) {
            hipLaunchKernelGGL(sum_over_slots_kernel, dim3((unsigned)((pc4 + 255) / 256)), dim3(256), 0, st,
                               (const float4*)dpre, (float4*)Dpart, N, (size_t)pc4);
            Gd = 1;
        }
    }
    if (C > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(l0_reduce_cls_kernel, dim3(N), dim3(4 * 3 * C), 0, st, rows, Rc, S, C);
    if (Dpart)
        hipLaunchKernelGGL(d_accumulate_kernel, dim3((pc4 + 255) / 256), dim3(256), 0, st, (const float4*)Dpart, Gd, pc4,
                           alpha, first, (float4*)Dacc);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// dz = Rc . Wcls^T, posterior gradients and the refinement "latent" vector
//   g_mean   = dz - mean
//   g_logvar = dz * 0.5 * exp(logvar/2) * eps - 0.5 * (exp(logvar) - 1)
//   latent   = [mean | logvar | LN(g_mean) | LN(g_logvar)]   (iodine.py:253-275; 3-D layernorm :382-384,394)
// One block (L threads rounded up to a wave multiple) per slot.
// -----------------------------------------------------------------------------------------------
IOD_DEVINL float block_sum_f(float v, float* s_buf, int tid, int nthreads)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) s_buf[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (nthreads + 63) / 64; ++w) t += s_buf[w];
    return t;
}

__global__ void dz_latent_kernel(const float* __restrict__ Rc, const float* __restrict__ wclsT /*[9][C][L]*/,
                                 const float* __restrict__ pm, const float* __restrict__ plv,
                                 const float* __restrict__ eps, int L, int C, int use_ln,
                                 float* __restrict__ g_pm, float* __restrict__ g_plv, float* __restrict__ latent, int Lreal)
{
    // Lreal <= L: the layer-norm of iodine.py:376-395 (3-D case: mean and UNBIASED std over the latent axis) runs over the first Lreal
    // entries - the reference's DIM_LATENT; entries Lreal .. L - 1 exist only when the host padded the latent axis to a multiple of 4
    // (zero weights, zero posterior: their gradients are exactly 0) and are written as 0
    // blockDim = NS * Lp (Lp = L rounded up to 64, NS = 8 slices for L <= 64): the 9C-long contraction is cut into NS slices (one per group of
    // Lp threads), eight independent partial sums each (eight L2 loads in flight per thread: the kernel is a chain of load latencies), combined
    // in fixed order; slice 0 finishes the row.  (One thread per latent walked all 9C terms as a single dependent load + fma chain: 64 us per
    // launch at cfg3; four slices x four partial sums: 21 us.)
    extern __shared__ float s_rc[];                     // 9*C, then NS*Lp partial sums
    __shared__ float s_buf[8];
    const int n = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const int Lp = (L + 63) / 64 * 64, NS = nth / Lp, l = tid % Lp, slice = tid / Lp, J = 9 * C;
    float* s_dz = s_rc + J;
    for (int i = tid; i < J; i += nth) s_rc[i] = Rc[(size_t)n * J + i];
    __syncthreads();
    {
        const int per = (J + NS - 1) / NS, j0 = slice * per, j1 = min(J, j0 + per);
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (l < L) {
            const float* w = wclsT + l;
            int j = j0;
            for (; j + 7 < j1; j += 8) {
                float wv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) wv[q] = w[(size_t)(j + q) * L];
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] = fmaf(s_rc[j + q], wv[q], a[q]);
            }
            for (; j < j1; ++j) a[0] = fmaf(s_rc[j], w[(size_t)j * L], a[0]);
        }
        s_dz[slice * Lp + l] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();
    const bool act = slice == 0 && l < L;
    float gm = 0.f, gl = 0.f, mu = 0.f, lv = 0.f;
    if (act) {
        float dz = 0.f;
        for (int q = 0; q < NS; ++q) dz += s_dz[q * Lp + l];
        mu = pm[(size_t)n * L + tid]; lv = plv[(size_t)n * L + tid];
        const float e = eps[(size_t)n * L + tid];
        gm = dz - mu;
        gl = dz * 0.5f * expf(0.5f * lv) * e - 0.5f * (expf(lv) - 1.f);
        g_pm[(size_t)n * L + tid] = gm;
        g_plv[(size_t)n * L + tid] = gl;
    }
    float nm = gm, nl = gl;
    if (use_ln) {
        const bool real = act && l < Lreal;
        const float m1 = block_sum_f(real ? gm : 0.f, s_buf, tid, nth) / Lreal;
        const float m2 = block_sum_f(real ? gl : 0.f, s_buf, tid, nth) / Lreal;
        const float d1 = real ? gm - m1 : 0.f, d2 = real ? gl - m2 : 0.f;
        const float v1 = block_sum_f(d1 * d1, s_buf, tid, nth) / (Lreal - 1);      // torch.std: unbiased
        const float v2 = block_sum_f(d2 * d2, s_buf, tid, nth) / (Lreal - 1);
        nm = real ? d1 / (sqrtf(v1) + 1e-5f) : 0.f;
        nl = real ? d2 / (sqrtf(v2) + 1e-5f) : 0.f;
    }
    if (act) {
        float* o = latent + (size_t)n * 4 * L;
        o[tid] = mu; o[L + tid] = lv; o[2 * L + tid] = nm; o[3 * L + tid] = nl;
    }
}

hipError_t launch_dz_latent(hipStream_t st, const float* Rc, const float* wclsT, const float* pm, const float* plv,
                            const float* eps, int N, int L, int C, int use_ln, float* g_pm, float* g_plv, float* latent, int Lreal)
{
    if (Lreal <= 0 || Lreal > L) Lreal = L;
    IOD_XSKIP(32);
    const int Lp = (L + 63) / 64 * 64;
    if (Lp > 512) return hipErrorInvalidValue;                   // block_sum_f: at most 8 waves
    const int nth = (512 / Lp) * Lp;                             // 8 slices for L <= 64, 4 for L <= 128, ...
    hipLaunchKernelGGL(dz_latent_kernel, dim3(N), dim3(nth), (9 * C + nth) * sizeof(float), st, Rc, wclsT, pm, plv, eps, L, C,
                       use_ln, g_pm, g_plv, latent, Lreal);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// KL against N(0,1) (iodine.py:653-659,191-193) + ELBO assembly.  One block per image; a second
// single-block kernel forms the batch means in fixed order.
//   img_terms[b] = {ll_b, kl_b};  scal = {elbo, kl, ll} (means over the B local images)
// -----------------------------------------------------------------------------------------------
__global__ void kl_image_kernel(const float* __restrict__ pm, const float* __restrict__ plv,
                                const float* __restrict__ ll_img, int KL_, float* __restrict__ img_terms)
{
    __shared__ float s_buf[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < KL_; i += blockDim.x) {
        const float mu = pm[(size_t)b * KL_ + i], lv = plv[(size_t)b * KL_ + i];
        s += 0.5f * (expf(lv) + mu * mu - 1.f - lv);
    }
    const float kl = block_sum_f(s, s_buf, tid, blockDim.x);
    if (tid == 0) { img_terms[2 * b] = ll_img[b]; img_terms[2 * b + 1] = kl; }
}

__global__ void elbo_mean_kernel(const float* __restrict__ img_terms, int B, float* __restrict__ scal)
{
    if (threadIdx.x == 0) {
        double ll = 0.0, kl = 0.0;
        for (int b = 0; b < B; ++b) { ll += img_terms[2 * b]; kl += img_terms[2 * b + 1]; }
        ll /= B; kl /= B;
        scal[0] = (float)(ll - kl); scal[1] = (float)kl; scal[2] = (float)ll;
    }
}

hipError_t launch_elbo(hipStream_t st, const float* pm, const float* plv, const float* ll_img, int B, int K, int L,
                       float* img_terms, float* scal)
{
    hipLaunchKernelGGL(kl_image_kernel, dim3(B), dim3(256), 0, st, pm, plv, ll_img, K * L, img_terms);
    hipLaunchKernelGGL(elbo_mean_kernel, dim3(1), dim3(64), 0, st, img_terms, B, scal);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// Refinement head (RefinementNetwork.forward after the conv stack, iodine.py:481-503):
//   avg-pool -> Linear(C->H) -> ELU(ELU(.)) -> [u | latent] -> LSTMCell -> updates read from the CELL state c1
//   (the reference's `(c, h) = lstm(...)` names h1 "c" and c1 "h") -> lambda += delta (iodine.py:642-643).
// One block per slot; weights pre-transposed to [in][out] so lane j streams column j coalesced.
//   saved (training): pooled[n][C], s[n][H] (MLP pre-activation), gates[n][4H] (post-activation i,f,g,o),
//   xin[n][H+4L] (LSTM input); the state buffers are per-iteration in training mode.
// -----------------------------------------------------------------------------------------------
// STAGE 0: everything in one launch (the gate pre-activations as sliced matrix-vector products: every block streams the 3.1 MB of LSTM weights
// from L2 - 0.74 GB per launch at cfg3, L2-bandwidth-bound at 55 us).  STAGE 1 / 2 (round 4): the kernel stops after assembling the LSTM input
// row xh[n] = [u | latent | h_prev] resp. resumes with the gate pre-activations gp[n][4H] that an fp32-MFMA GEMM over all slots produced in
// between (launch_sgemm_tn_mfma, A row-major: the weights are read once per 32 slots).
template <int STAGE>
__global__ __launch_bounds__(1024)
void refine_head_kernel(const float* __restrict__ feat /*[N][PL][C]*/, int PL, int C, int H, int L,
                        const float* __restrict__ mlp_wT /*[C][H]*/, const float* __restrict__ mlp_b,
                        const float* __restrict__ wihT /*[H+4L][4H]*/, const float* __restrict__ whhT /*[H][4H]*/,
                        const float* __restrict__ lstm_b /*[4H] = b_ih + b_hh*/,
                        const float* __restrict__ wmT /*[H][L]*/, const float* __restrict__ bm,
                        const float* __restrict__ wvT /*[H][L]*/, const float* __restrict__ bv,
                        const float* __restrict__ latent /*[N][4L]*/,
                        const float* __restrict__ h_prev, const float* __restrict__ c_prev,
                        float* __restrict__ h_out, float* __restrict__ c_out,
                        float* __restrict__ pm, float* __restrict__ plv,
                        float* __restrict__ sv_pooled, float* __restrict__ sv_s, float* __restrict__ sv_gates,
                        float* __restrict__ sv_xin, float* __restrict__ d_mean_out, float* __restrict__ d_logvar_out,
                        float* __restrict__ xh /*[.][H + 4L + H]*/, const float* __restrict__ gp /*[.][4H]*/)
{
    extern __shared__ float sm[];
    float* s_pool = sm;                 // C
    float* s_x = s_pool + C;            // H + 4L
    float* s_h = s_x + H + 4 * L;       // H   (h_prev)
    float* s_c = s_h + H;               // H   (c1)
    float* s_red = s_c + H;             // 256
    float* s_part = s_red + 256;        // [16][4H]: LSTM partial gate sums of the K slices
    // 1024 threads: the LSTM reduction (H + 4L + H = 768 terms per gate) is split over 4 K slices (tid >> 8); everything
    // else runs on the first 256 threads.  One block per slot: the serial depth of the gate loop set this kernel's time.
    const int n = blockIdx.x, tid = threadIdx.x & 255, ksl = threadIdx.x >> 8;
    const bool lead = ksl == 0;
    // The matrix-vector products below (MLP, LSTM gates, posterior update) stream 3.3 MB of weights through this CU.  With
    // one output column per thread that is a dword load per term and one L2 round trip per fma; instead every thread owns
    // four adjacent columns (16-byte loads at the full rate of the texture path), the rows are cut into up to 16 slices
    // that run side by side, and a few rows' weights are requested before the first fma.
    // part[ks][col] = sum over rows r = ks, ks + NS, ... of v[r] * W[r][col]: four adjacent columns per thread (16-byte
    // loads), NS = min(16, 1024 / (ncols / 4)) row slices, UR rows' weights requested before the first fma.  The caller
    // adds the NS partial rows in slice order after a barrier.  Returns NS.
    auto matvec4 = [&](const float* __restrict__ v, int nrows, const float* __restrict__ W, int ld, int ncols,
                       float* __restrict__ part) -> int {
        const int CQ = ncols / 4, NS = min(16, 1024 / CQ);
        const int t = threadIdx.x, cq = t % CQ, ks = t / CQ;
        if (ks < NS) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            constexpr int UR = 4;
            for (int r0 = ks; r0 < nrows; r0 += NS * UR) {
                float4 w[UR];
                float xv[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const int r = r0 + u * NS, rr = r < nrows ? r : ks;
                    w[u] = *reinterpret_cast<const float4*>(W + (size_t)rr * ld + 4 * cq);
                    xv[u] = r < nrows ? v[rr] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    acc.x = fmaf(xv[u], w[u].x, acc.x); acc.y = fmaf(xv[u], w[u].y, acc.y);
                    acc.z = fmaf(xv[u], w[u].z, acc.z); acc.w = fmaf(xv[u], w[u].w, acc.w);
                }
            }
            *reinterpret_cast<float4*>(part + (size_t)ks * ncols + 4 * cq) = acc;
        }
        return NS;
    };

    // global average pool over the PL pixels of the last conv layer (F.adaptive_avg_pool2d, iodine.py:481)
    if constexpr (STAGE != 2) {
        const int c = tid % C, g = tid / C, G = 256 / C;
        float s = 0.f;
        if (lead) {
            for (int p = g; p < PL; p += G) s += feat[((size_t)n * PL + p) * C + c];
            s_red[tid] = s;
        }
        __syncthreads();
        if (lead && tid < C) {
            float t = 0.f;
            for (int j = 0; j < G; ++j) t += s_red[j * C + tid];
            t /= (float)PL;
            s_pool[tid] = t;
            if (sv_pooled) sv_pooled[(size_t)n * C + tid] = t;
        }
        __syncthreads();
    }
    // MLP + double ELU
    const int IN = H + 4 * L, H4 = 4 * H;
    if constexpr (STAGE != 2) {
        const int ns_mlp = matvec4(s_pool, C, mlp_wT, H, H, s_part);
        __syncthreads();
        for (int j = tid; lead && j < H; j += 256) {
            float s = mlp_b[j];
            for (int q = 0; q < ns_mlp; ++q) s += s_part[(size_t)q * H + j];
            const float u = elu1(elu1(s));
            s_x[j] = u;
            if (sv_s) sv_s[(size_t)n * H + j] = s;               // pre-activation (training backward recomputes the ELUs)
            s_h[j] = h_prev[(size_t)n * H + j];
        }
        for (int j = tid; lead && j < 4 * L; j += 256) s_x[H + j] = latent[(size_t)n * 4 * L + j];
        __syncthreads();
        if (sv_xin && lead)
            for (int j = tid; j < H + 4 * L; j += 256) sv_xin[(size_t)n * (H + 4 * L) + j] = s_x[j];
    }
    if constexpr (STAGE == 1) {                                  // the row of the gate GEMM: [u | latent | h_prev]
        for (int j = threadIdx.x; j < IN + H; j += 1024) xh[(size_t)n * (IN + H) + j] = j < IN ? s_x[j] : s_h[j - IN];
        return;
    }
    // LSTM cell, gate order i, f, g, o (torch.nn.LSTMCell)
    // Gate pre-activations: the (H + 4L + H)-term contraction is cut into NS = 16 K slices (rows r = ks, ks + NS, ... of
    // [W_ih^T ; W_hh^T]) and every thread owns FOUR adjacent gate columns, so a lane moves 16 bytes per load: with one
    // column per thread the 3.2 MB of head weights went through the CU's texture path as dword loads (~0.1 ms per launch).
    // Partial sums meet in LDS and are added in slice order (deterministic, independent of the slot's position).
    if constexpr (STAGE == 0) {
        const int HQ = H / 4;                                    // column groups
        const int NS = min(16, 1024 / HQ);
        const int t = threadIdx.x, jq = t % HQ, ks = t / HQ;
        if (ks < NS) {
            float4 gi = make_float4(0.f, 0.f, 0.f, 0.f), gf = gi, gg = gi, go = gi;
            const int R = IN + H;
            constexpr int UR = 4;
            for (int r0 = ks; r0 < R; r0 += NS * UR) {
                float4 w[UR][4];
                float xv[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const int r = r0 + u * NS;
                    const int rr = r < R ? r : ks;
                    const float* wrow = (rr < IN ? wihT + (size_t)rr * H4 : whhT + (size_t)(rr - IN) * H4) + 4 * jq;
                    w[u][0] = *reinterpret_cast<const float4*>(wrow);
                    w[u][1] = *reinterpret_cast<const float4*>(wrow + H);
                    w[u][2] = *reinterpret_cast<const float4*>(wrow + 2 * H);
                    w[u][3] = *reinterpret_cast<const float4*>(wrow + 3 * H);
                    xv[u] = r < R ? (rr < IN ? s_x[rr] : s_h[rr - IN]) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const float x = xv[u];
                    gi.x = fmaf(x, w[u][0].x, gi.x); gi.y = fmaf(x, w[u][0].y, gi.y); gi.z = fmaf(x, w[u][0].z, gi.z); gi.w = fmaf(x, w[u][0].w, gi.w);
                    gf.x = fmaf(x, w[u][1].x, gf.x); gf.y = fmaf(x, w[u][1].y, gf.y); gf.z = fmaf(x, w[u][1].z, gf.z); gf.w = fmaf(x, w[u][1].w, gf.w);
                    gg.x = fmaf(x, w[u][2].x, gg.x); gg.y = fmaf(x, w[u][2].y, gg.y); gg.z = fmaf(x, w[u][2].z, gg.z); gg.w = fmaf(x, w[u][2].w, gg.w);
                    go.x = fmaf(x, w[u][3].x, go.x); go.y = fmaf(x, w[u][3].y, go.y); go.z = fmaf(x, w[u][3].z, go.z); go.w = fmaf(x, w[u][3].w, go.w);
                }
            }
            float* ps = s_part + (size_t)ks * H4 + 4 * jq;
            *reinterpret_cast<float4*>(ps) = gi;
            *reinterpret_cast<float4*>(ps + H) = gf;
            *reinterpret_cast<float4*>(ps + 2 * H) = gg;
            *reinterpret_cast<float4*>(ps + 3 * H) = go;
        }
    }
    __syncthreads();
    for (int j = tid; lead && j < H; j += 256) {
        float gi = lstm_b[j], gf = lstm_b[H + j], gg = lstm_b[2 * H + j], go = lstm_b[3 * H + j];
        if constexpr (STAGE == 2) {
            const float* ps = gp + (size_t)n * H4;
            gi += ps[j]; gf += ps[H + j]; gg += ps[2 * H + j]; go += ps[3 * H + j];
        } else {
            const int NS = min(16, 1024 / (H / 4));
            for (int q = 0; q < NS; ++q) {
                const float* ps = s_part + (size_t)q * H4;
                gi += ps[j]; gf += ps[H + j]; gg += ps[2 * H + j]; go += ps[3 * H + j];
            }
        }
        gi = sigmoidf_(gi); gf = sigmoidf_(gf); gg = tanhf(gg); go = sigmoidf_(go);
        const float c1 = gf * c_prev[(size_t)n * H + j] + gi * gg;
        const float h1 = go * tanhf(c1);
        s_c[j] = c1;
        c_out[(size_t)n * H + j] = c1;
        h_out[(size_t)n * H + j] = h1;
        if (sv_gates) {
            float* gs = sv_gates + (size_t)n * H4;
            gs[j] = gi; gs[H + j] = gf; gs[2 * H + j] = gg; gs[3 * H + j] = go;
        }
    }
    __syncthreads();
    // posterior update from the cell state
    const int ns_up = matvec4(s_c, H, wmT, L, L, s_part);
    matvec4(s_c, H, wvT, L, L, s_part + 16 * L);
    __syncthreads();
    for (int t = tid; lead && t < 2 * L; t += 256) {
        const int l = t % L;
        const bool is_lv = t >= L;
        const float* pp = s_part + (is_lv ? 16 * L : 0);
        float s = is_lv ? bv[l] : bm[l];
        for (int q = 0; q < ns_up; ++q) s += pp[(size_t)q * L + l];
        float* dst = (is_lv ? plv : pm) + (size_t)n * L + l;
        *dst = *dst + s;
        float* dd = is_lv ? d_logvar_out : d_mean_out;
        if (dd) dd[(size_t)n * L + l] = s;
    }
}

// xh / gp (both or neither): scratch [roundup32(N)][H + 4L + H] / [roundup32(N)][4H] for the three-launch form (pre, gate GEMM on fp32 MFMA,
// post); wihT must then be followed in memory by whhT ([H + 4L + H][4H] contiguous: the GEMM's B operand)
hipError_t launch_refine_head(hipStream_t st, const float* feat, int N, int PL, int C, int H, int L,
                              const float* mlp_wT, const float* mlp_b, const float* wihT, const float* whhT,
                              const float* lstm_b, const float* wmT, const float* bm, const float* wvT, const float* bv,
                              const float* latent, const float* h_prev, const float* c_prev, float* h_out, float* c_out,
                              float* pm, float* plv, float* sv_pooled, float* sv_s, float* sv_gates, float* sv_xin,
                              float* d_mean, float* d_logvar, float* xh, float* gp)
{
    IOD_XSKIP(4);
    if (256 % C != 0) return hipErrorInvalidValue;
    if ((H + 4 * L) % 4 != 0 || H % 4 != 0 || L % 4 != 0 || C % 4 != 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)(C + (H + 4 * L) + H + H + 256 + 16 * 4 * H) * sizeof(float);
    static std::atomic<unsigned> attr_devs[3];                             // devices each instance is configured on
    if (lds > 96 * 1024) return hipErrorInvalidValue;
#define HEAD_LAUNCH(STG) do {                                                                                                        \
        if (hipError_t e = iod_set_max_lds((const void*)refine_head_kernel<STG>, 96 * 1024, attr_devs[STG]); e != hipSuccess) return e;  \
        hipLaunchKernelGGL(refine_head_kernel<STG>, dim3(N), dim3(1024), lds, st, feat, PL, C, H, L, mlp_wT, mlp_b, wihT, whhT,      \
                           lstm_b, wmT, bm, wvT, bv, latent, h_prev, c_prev, h_out, c_out, pm, plv, sv_pooled, sv_s,                 \
                           sv_gates, sv_xin, d_mean, d_logvar, xh, gp);                                                              \
    } while (0)
    const int Kg = H + 4 * L + H, Np = (N + 31) / 32 * 32;
    if (xh && gp && whhT == wihT + (size_t)(H + 4 * L) * 4 * H && sgemm_tn_mfma_ok(Np, 4 * H, Kg)) {
        HEAD_LAUNCH(1);
        if (hipError_t e = launch_sgemm_tn_mfma(st, Np, 4 * H, Kg, 1.f, xh, Kg, wihT, 4 * H, 0.f, gp, 4 * H, 0, 0, 1); e != hipSuccess) return e;
        HEAD_LAUNCH(2);
    } else {
        HEAD_LAUNCH(0);
    }
#undef HEAD_LAUNCH
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// small utilities: transpose [R][Cc] -> [Cc][R], vector add
// -----------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc)
{
    const size_t total = (size_t)R * Cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = i / Cc, c = i % Cc;
        dst[(size_t)c * R + r] = src[i];
    }
}

__global__ void multi_copy_kernel(const MultiCopy mc)
{
    const int j = blockIdx.y;
    const float* __restrict__ src = mc.src[j];
    float* __restrict__ dst = mc.dst[j];
    const int op = mc.op[j], n = mc.n[j];
    if (op == 1) {                                           // transpose [R][Cc] -> [Cc][R]
        const int Cc = mc.cols[j], R = n / Cc;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[(size_t)(i % Cc) * R + i / Cc] = src[i];
    } else if (op == 2) {
        const float* __restrict__ s2 = mc.src2[j];
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i] + s2[i];
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
    }
}

hipError_t launch_multi_copy(hipStream_t st, const MultiCopy& mc)
{
    if (mc.count <= 0) return hipSuccess;
    int nmax = 1;
    for (int j = 0; j < mc.count; ++j) nmax = mc.n[j] > nmax ? mc.n[j] : nmax;
    const int bx = (nmax + 255) / 256 < 256 ? (nmax + 255) / 256 : 256;
    hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, mc.count), dim3(256), 0, st, mc);
    return hipGetLastError();
}

hipError_t launch_transpose(hipStream_t st, const float* src, float* dst, int R, int Cc)
{
    IOD_XSKIP(64);
    const int blocks = (int)std::min<size_t>(((size_t)R * Cc + 255) / 256, 2048);
    hipLaunchKernelGGL(transpose_kernel, dim3(blocks), dim3(256), 0, st, src, dst, R, Cc);
    return hipGetLastError();
}

__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

hipError_t launch_add2(hipStream_t st, const float* a, const float* b, float* o, int n)
{
    hipLaunchKernelGGL(add2_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, o, n);
    return hipGetLastError();
}

// decoder output conv weights OIHW [4][C][3][3] -> [tap][ci][4]
__global__ void pack_dec_out_kernel(const float* __restrict__ w, float* __restrict__ wk, int C)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * C * 4) return;
    const int co = i & 3, ci = (i >> 2) % C, tap = (i >> 2) / C;
    wk[i] = w[((size_t)co * C + ci) * 9 + tap];
}

hipError_t launch_pack_dec_out(hipStream_t st, const float* w, float* wk, int C)
{
    hipLaunchKernelGGL(pack_dec_out_kernel, dim3((9 * C * 4 + 255) / 256), dim3(256), 0, st, w, wk, C);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// the optimizer step adjacent to the training step (lib/solver/build.py:5-16 builds
// torch.optim.Adam over every parameter; lib/engine/train.py:65 calls .step()).  One launch over all tensors:
// ptrs[4*t + {0,1,2,3}] = {param, grad, exp_avg, exp_avg_sq} of tensor t, offs[t] = first flat element of tensor t.
// Arithmetic follows torch.optim.Adam (amsgrad=False, maximize=False): weight decay added to the gradient,
// bias corrections 1 - beta^step, denom = sqrt(v) / sqrt(bc2) + eps, p -= (lr / bc1) * m / denom.
// -----------------------------------------------------------------------------------------------
__global__ void adam_multi_kernel(const long long* __restrict__ ptrs, const long long* __restrict__ offs, int n_tensors,
                                  long long total, float step_size, float beta1, float beta2, float omb1, float omb2,
                                  float eps, float wd, float bc2_sqrt)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_tensors - 1;                       // last tensor whose offset <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offs[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const long long j = i - offs[lo];
        float* p = reinterpret_cast<float*>(ptrs[4 * lo + 0]);
        const float* g = reinterpret_cast<const float*>(ptrs[4 * lo + 1]);
        float* m = reinterpret_cast<float*>(ptrs[4 * lo + 2]);
        float* v = reinterpret_cast<float*>(ptrs[4 * lo + 3]);
        float grad = g[j];
        const float pv = p[j];
        if (wd != 0.f) grad = fmaf(wd, pv, grad);
        const float mn = fmaf(omb1, grad, beta1 * m[j]);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vn = fmaf(omb2 * grad, grad, beta2 * v[j]);   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[j] = mn; v[j] = vn;
        const float denom = sqrtf(vn) / bc2_sqrt + eps;
        p[j] = pv - step_size * (mn / denom);
    }
}

hipError_t launch_adam_multi(hipStream_t st, const long long* ptrs, const long long* offs, int n_tensors, long long total,
                             double lr, double beta1, double beta2, double eps, double wd, int step)
{
    // scalar prefactors in double on the host, like the Python floats torch.optim.Adam derives them from
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2_sqrt = sqrt(1.0 - pow(beta2, (double)step));
    const int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, st, ptrs, offs, n_tensors, total, (float)(lr / bc1),
                       (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)wd,
                       (float)bc2_sqrt);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// ARI evaluation epilogue of reconstruct (lib/eval/ari_eval.py:25-39, lib/utils/ari.py:36-52):
// argmax over the K slot masks per pixel, then the contingency table  table[b][i][k] = #pixels with gt_i set and
// argmax == k  (integer arithmetic, int32 atomics on an LDS copy of the table, one flush per block).
//   mask (B, K, 1, P) fp32 as returned by reconstruct; gt (B, G, P) uint8 0/1 (padded with empty masks to G rows).
// -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void ari_table_kernel(const float* __restrict__ mask, const unsigned char* __restrict__ gt, int K, int G, int P,
                      int* __restrict__ table)
{
    extern __shared__ int s_tab[];                            // G*K
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < G * K; i += 256) s_tab[i] = 0;
    __syncthreads();
    const float* mb = mask + (size_t)b * K * P;
    const unsigned char* gb = gt + (size_t)b * G * P;
    for (int p = blockIdx.x * 256 + tid; p < P; p += gridDim.x * 256) {
        int best = 0;
        float bv = mb[p];
        for (int k = 1; k < K; ++k) {                         // torch.argmax: first maximal index
            const float v = mb[(size_t)k * P + p];
            if (v > bv) { bv = v; best = k; }
        }
        for (int i = 0; i < G; ++i)
            if (gb[(size_t)i * P + p]) atomicAdd(&s_tab[i * K + best], 1);
    }
    __syncthreads();
    for (int i = tid; i < G * K; i += 256)
        if (s_tab[i]) atomicAdd(&table[(size_t)b * G * K + i], s_tab[i]);
}

hipError_t launch_ari_table(hipStream_t st, const float* mask, const unsigned char* gt, int B, int K, int G, int P,
                            int* table)
{
    hipError_t e = hipMemsetAsync(table, 0, sizeof(int) * (size_t)B * G * K, st);
    if (e != hipSuccess) return e;
    const int bx = std::min((P + 255) / 256, 16);
    hipLaunchKernelGGL(ari_table_kernel, dim3(bx, B), dim3(256), (size_t)G * K * sizeof(int), st, mask, gt, K, G, P, table);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// Counter-based standard normals for Gaussian.sample (iodine.py:620-634 calls torch.randn_like; the values of a torch
// generator are not part of the reference's contract, only their distribution).  Philox4x32-10 (Salmon et al., SC'11):
// counter = {quad index lo, hi, stream lo, hi}, key = {seed lo, hi}; the four 32-bit outputs of one call give two
// Box-Muller pairs.  Element e of the output is normal #(e & 3) of quad e >> 2, so any sub-range / any launch shape
// produces the same numbers (oracle/philox_oracle.py restates this in numpy; tests/test_gpu_extras.py compares).
// -----------------------------------------------------------------------------------------------
IOD_DEVINL void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256)
void randn_philox_kernel(float* __restrict__ out, long long n, unsigned k0, unsigned k1, unsigned s0, unsigned s1)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= n) return;
    unsigned r[4];
    philox4x32_10((unsigned)q, (unsigned)((unsigned long long)q >> 32), s0, s1, k0, k1, r);
    float v[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float u1 = ((float)(r[2 * j] >> 9) + 0.5f) * 1.1920928955078125e-07f;        // (0, 1), exact in fp32
        const float u2 = ((float)(r[2 * j + 1] >> 9) + 0.5f) * 1.1920928955078125e-07f;
        const float rad = sqrtf(-2.f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        v[2 * j] = rad * cs; v[2 * j + 1] = rad * sn;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (q * 4 + j < n) out[q * 4 + j] = v[j];
}

hipError_t launch_randn_philox(hipStream_t st, float* out, long long n, unsigned long long seed, unsigned long long stream_id)
{
    const long long quads = (n + 3) / 4;
    hipLaunchKernelGGL(randn_philox_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, out, n,
                       (unsigned)seed, (unsigned)(seed >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32));
    return hipGetLastError();
}
