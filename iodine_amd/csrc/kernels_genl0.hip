// Generic decoder, layer 0 (SpatialBroadcast + the first conv of MultiLayerConv, iodine.py:505-540, 570-594) WITHOUT the broadcast tensor.
// Round 5.  Until now the generic path (any odd DEC.KERNEL_SIZE / channel count, kernels_generic.hip) materialised [N][P][L + 2] and ran
// layer 0 like any other conv: at the reference's default arch (KERNEL_SIZE 5, lib/config/defaults.py:100; CLEVR shapes, 28 slot-images)
// 1.0 ms forward, 1.0 ms data gradient, 2.6 ms weight gradient and two pixel-sum launches per decoder pass - a quarter of the training step -
// for an input whose L latent channels are CONSTANT over the image.  With that:
//   forward    pre[n][y][x][co] = cterm[y][x][co] + sum_{taps valid at (y, x)} U[n][tap][co],   U[n][tap][co] = sum_ci w[tap][ci][co] z[n][ci]
//              cterm = bias + the conv of the two coordinate channels (the same for every slot-image: built once per set_params);
//              the valid taps of a pixel are a rectangle [ky0, ky1] x [kx0, kx1] (zero padding), so the sum is four reads of the 2-D
//              prefix table PS[n][a][b][co] = sum_{ky < a, kx < b} U[n][ky][kx][co];
//   backward   T[n][tap][co] = sum_{p : p + tap inside} dpre[n][p][co]                       (tap-window sums of the gradient)
//              gw[co][ci][tap] += alpha sum_n z[n][ci] T[n][tap][co]  (ci < L),   gb[co] += alpha sum_n T[n][centre][co],
//              dz[n][ci] = sum_{tap, co} w[tap][ci][co] T[n][tap][co];
//              coordinate channels: TX[n][tap][co] = sum_p lin[x + kx - PAD] dpre (same window), TY likewise with lin[y + ky - PAD].
// One pass over dpre (per image row: the full row sum, the k windowed row sums = full minus at most PAD edge pixels, and the k
// x-coordinate-weighted sums), then per-tap sums over the rows.  Everything is plain fp32 in a fixed order (deterministic); the sums differ
// from the materialised conv's only in rounding (a different association of the same terms).  Any odd k <= GEN_L0_KMAX, any L, Co, S.
#include "common.h"

namespace {

constexpr int KMAX = GEN_L0_KMAX;

IOD_DEVINL float l0_elu(float v) { return v > 0.f ? v : expm1f(v); }

// cterm[p][co] = bias[co] + sum_{valid taps} (wt[tap][L][co] lin[x + kx - PAD] + wt[tap][L + 1][co] lin[y + ky - PAD])
__global__ void gen_l0_coord_kernel(const float* __restrict__ wt, const float* __restrict__ bias, const float* __restrict__ lin, int L, int S,
                                    int Co, int k, float* __restrict__ cterm)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)S * S * Co) return;
    const int co = (int)(idx % Co), p = (int)(idx / Co), y = p / S, x = p % S, pad = k / 2, Ci = L + 2;
    float v = bias[co];
    for (int ky = 0; ky < k; ++ky) {
        const int iy = y + ky - pad;
        if ((unsigned)iy >= (unsigned)S) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = x + kx - pad;
            if ((unsigned)ix >= (unsigned)S) continue;
            const float* w = wt + ((size_t)(ky * k + kx) * Ci + L) * Co + co;
            v = fmaf(w[0], lin[ix], v);
            v = fmaf(w[Co], lin[iy], v);
        }
    }
    cterm[idx] = v;
}

// U[n][tap][co] = sum_ci wt[tap][ci][co] z[n][ci]: one thread per output, two partial sums
__global__ __launch_bounds__(256) void gen_l0_u_kernel(const float* __restrict__ z, const float* __restrict__ wt, int N, int L, int Co, int k,
                                                       float* __restrict__ u)
{
    const int KK = k * k, Ci = L + 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * KK * Co) return;
    const int co = (int)(idx % Co), tap = (int)((idx / Co) % KK);
    const float* zn = z + (idx / ((size_t)Co * KK)) * L;
    const float* w = wt + (size_t)tap * Ci * Co + co;
    float s0 = 0.f, s1 = 0.f;
    int ci = 0;
    for (; ci + 1 < L; ci += 2) { s0 = fmaf(w[(size_t)ci * Co], zn[ci], s0); s1 = fmaf(w[(size_t)(ci + 1) * Co], zn[ci + 1], s1); }
    if (ci < L) s0 = fmaf(w[(size_t)ci * Co], zn[ci], s0);
    u[idx] = s0 + s1;
}

// PS[n][a][b][co] = sum_{ky < a, kx < b} U[n][ky][kx][co], a, b = 0 .. k: one thread per (n, co)
__global__ __launch_bounds__(256) void gen_l0_prefix_kernel(const float* __restrict__ u, int N, int Co, int k, float* __restrict__ ps)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, k1 = k + 1;
    if (idx >= N * Co) return;
    const int n = idx / Co, co = idx % Co;
    const float* un = u + (size_t)n * k * k * Co;
    float* pn = ps + (size_t)n * k1 * k1 * Co;
    float prev[KMAX];                                          // PS row a - 1, columns 1 .. k
#pragma unroll
    for (int b = 0; b < KMAX; ++b) prev[b] = 0.f;
    for (int b = 0; b <= k; ++b) pn[(size_t)b * Co + co] = 0.f;
    for (int a = 1; a <= k; ++a) {
        pn[(size_t)a * k1 * Co + co] = 0.f;
        float row = 0.f;
#pragma unroll
        for (int b = 1; b <= KMAX; ++b) {
            if (b > k) break;
            row += un[(size_t)((a - 1) * k + b - 1) * Co + co];
            prev[b - 1] += row;
            pn[((size_t)a * k1 + b) * Co + co] = prev[b - 1];
        }
    }
}

// out[n][p][co .. co + V) = elu(cterm[p][..] + window sum of PS[n])
template <int V>
__global__ __launch_bounds__(256) void gen_l0_fwd_kernel(const float* __restrict__ ps, const float* __restrict__ cterm, int S, int Co, int k,
                                                         size_t total, float* __restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = Co / V, k1 = k + 1, pad = k / 2;
    const int co = (int)(idx % cv) * V;
    const size_t px = idx / cv;
    const int p = (int)(px % ((size_t)S * S)), y = p / S, x = p % S;
    const size_t n = px / ((size_t)S * S);
    const int a0 = max(0, pad - y), a1 = min(k - 1, S - 1 + pad - y) + 1, b0 = max(0, pad - x), b1 = min(k - 1, S - 1 + pad - x) + 1;
    const float* pn = ps + n * k1 * k1 * Co + co;
    const float* q11 = pn + ((size_t)a1 * k1 + b1) * Co;
    const float* q01 = pn + ((size_t)a0 * k1 + b1) * Co;
    const float* q10 = pn + ((size_t)a1 * k1 + b0) * Co;
    const float* q00 = pn + ((size_t)a0 * k1 + b0) * Co;
    const float* ct = cterm + (size_t)p * Co + co;
    float* o = out + px * Co + co;
    if constexpr (V == 4) {
        const float4 c = *reinterpret_cast<const float4*>(ct), s11 = *reinterpret_cast<const float4*>(q11), s01 = *reinterpret_cast<const float4*>(q01),
                     s10 = *reinterpret_cast<const float4*>(q10), s00 = *reinterpret_cast<const float4*>(q00);
        float4 r;
        r.x = l0_elu(c.x + ((s11.x - s01.x) - (s10.x - s00.x)));
        r.y = l0_elu(c.y + ((s11.y - s01.y) - (s10.y - s00.y)));
        r.z = l0_elu(c.z + ((s11.z - s01.z) - (s10.z - s00.z)));
        r.w = l0_elu(c.w + ((s11.w - s01.w) - (s10.w - s00.w)));
        *reinterpret_cast<float4*>(o) = r;
    } else {
        o[0] = l0_elu(ct[0] + ((q11[0] - q01[0]) - (q10[0] - q00[0])));
    }
}

// one block per (slot-image, image row): rows[n][y][kx][co] = sum_{x in X(kx)} d[n][y][x][co],  rows[n][y][k + kx][co] = the same sum weighted
// with lin[x + kx - PAD];  X(kx) = the pixels whose tap kx stays inside the row.  Threads = (channel, pixel sub-sequence); the partial sums of
// a channel are added in a fixed order.
__global__ __launch_bounds__(256) void gen_l0_rows_kernel(const float* __restrict__ d, const float* __restrict__ lin, int S, int Co, int k,
                                                          float* __restrict__ rows)
{
    __shared__ float red[KMAX + 1][256];
    extern __shared__ float s_lp[];                            // [S + 2 PAD]: lin with zero margins
    const int pad = k / 2, tid = threadIdx.x;
    const size_t row = blockIdx.x;                             // n * S + y
    const float* dr = d + row * S * (size_t)Co;
    float* ro = rows + row * 2 * k * (size_t)Co;
    for (int i = tid; i < S + 2 * pad; i += 256) s_lp[i] = (i >= pad && i < S + pad) ? lin[i - pad] : 0.f;
    __syncthreads();
    for (int c0 = 0; c0 < Co; c0 += 256) {
        const int cw = min(256, Co - c0), nx = 256 / cw;
        const int c = tid % cw, xs = tid / cw;
        float full = 0.f, wx[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) wx[j] = 0.f;
        if (xs < nx)
            for (int x = xs; x < S; x += nx) {
                const float v = dr[(size_t)x * Co + c0 + c];
                full += v;
#pragma unroll
                for (int j = 0; j < KMAX; ++j) if (j < k) wx[j] = fmaf(s_lp[x + j], v, wx[j]);
            }
        red[KMAX][tid] = full;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) red[j][tid] = wx[j];
        __syncthreads();
        for (int item = tid; item < k * cw; item += 256) {
            const int kx = item / cw, cc = item % cw;
            float f = 0.f, w = 0.f;
            for (int g = 0; g < nx; ++g) { f += red[KMAX][g * cw + cc]; w += red[kx][g * cw + cc]; }
            // the window of tap kx leaves out the first PAD - kx (kx < PAD) or the last kx - PAD (kx > PAD) pixels of the row
            float edge = 0.f;
            if (kx < pad) for (int x = 0; x < min(S, pad - kx); ++x) edge += dr[(size_t)x * Co + c0 + cc];
            if (kx > pad) for (int x = max(0, S - (kx - pad)); x < S; ++x) edge += dr[(size_t)x * Co + c0 + cc];
            ro[(size_t)kx * Co + c0 + cc] = f - edge;
            ro[(size_t)(k + kx) * Co + c0 + cc] = w;
        }
        __syncthreads();
    }
}

// T[0][n][tap][co] = sum_{y in Y(ky)} rows[n][y][kx][co], T[1] = the x-weighted sums likewise, T[2] = sum_y lin[y + ky - PAD] rows[n][y][kx][co]
__global__ __launch_bounds__(256) void gen_l0_taps_kernel(const float* __restrict__ rows, const float* __restrict__ lin, int N, int S, int Co, int k,
                                                          float* __restrict__ T)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, tot = (size_t)N * k * k * Co;
    if (idx >= tot) return;
    const int co = (int)(idx % Co), tap = (int)((idx / Co) % (k * k)), ky = tap / k, kx = tap % k, pad = k / 2;
    const size_t n = idx / ((size_t)Co * k * k);
    const int y0 = max(0, pad - ky), y1 = min(S, S + pad - ky);
    const float* r = rows + (n * S * 2 * k + kx) * Co + co;
    float t0 = 0.f, t1 = 0.f, tx0 = 0.f, tx1 = 0.f, ty0 = 0.f, ty1 = 0.f;
    int y = y0;
    for (; y + 1 < y1; y += 2) {
        const float a = r[(size_t)y * 2 * k * Co], b = r[(size_t)(y + 1) * 2 * k * Co];
        t0 += a; t1 += b;
        tx0 += r[((size_t)y * 2 + 1) * k * Co]; tx1 += r[((size_t)(y + 1) * 2 + 1) * k * Co];
        ty0 = fmaf(lin[y + ky - pad], a, ty0); ty1 = fmaf(lin[y + 1 + ky - pad], b, ty1);
    }
    if (y < y1) {
        const float a = r[(size_t)y * 2 * k * Co];
        t0 += a; tx0 += r[((size_t)y * 2 + 1) * k * Co]; ty0 = fmaf(lin[y + ky - pad], a, ty0);
    }
    T[idx] = t0 + t1; T[tot + idx] = tx0 + tx1; T[2 * tot + idx] = ty0 + ty1;
}

// gw[co][ci][tap] += alpha sum_n (ci < L: z[n][ci] T[0][n][tap][co]; ci = L: T[1]; ci = L + 1: T[2]);  gb[co] += alpha sum_n T[0][n][centre][co]
__global__ __launch_bounds__(256) void gen_l0_wgrad_kernel(const float* __restrict__ T, const float* __restrict__ z, int N, int L, int Co, int k,
                                                           float alpha, float* __restrict__ gw, float* __restrict__ gb)
{
    const int KK = k * k, Ci = L + 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nw = (size_t)KK * Ci * Co, tot = (size_t)N * KK * Co;
    if (idx >= nw + Co) return;
    if (idx >= nw) {
        const int co = (int)(idx - nw);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += T[((size_t)n * KK + (KK / 2)) * Co + co];
        gb[co] += alpha * s;
        return;
    }
    const int co = (int)(idx % Co), ci = (int)((idx / Co) % Ci), tap = (int)(idx / ((size_t)Co * Ci));
    const float* t = T + (ci < L ? 0 : (size_t)(ci - L + 1) * tot) + (size_t)tap * Co + co;
    float s = 0.f;
    if (ci < L) for (int n = 0; n < N; ++n) s = fmaf(z[(size_t)n * L + ci], t[(size_t)n * KK * Co], s);
    else for (int n = 0; n < N; ++n) s += t[(size_t)n * KK * Co];
    gw[((size_t)co * Ci + ci) * KK + tap] += alpha * s;
}

// dz[n][ci] = sum_{tap, co} wt[tap][ci][co] T[0][n][tap][co]: one wave per (n, ci), lanes over (tap, co), xor-shuffle tree (fixed order)
__global__ __launch_bounds__(64) void gen_l0_dz_kernel(const float* __restrict__ T, const float* __restrict__ wt, int L, int Co, int k, int ld,
                                                       float* __restrict__ dz)
{
    const int n = blockIdx.x / L, ci = blockIdx.x % L, KK = k * k, Ci = L + 2;
    const float* t = T + (size_t)n * KK * Co;
    float s = 0.f;
    for (int e = threadIdx.x; e < KK * Co; e += 64) {
        const int tap = e / Co, co = e % Co;
        s = fmaf(wt[((size_t)tap * Ci + ci) * Co + co], t[e], s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) dz[(size_t)n * ld + ci] = s;
}

inline unsigned nblk(size_t total) { return (unsigned)((total + 255) / 256); }

} // namespace

size_t gen_l0_scratch_floats(int N, int S, int Co, int k)
{
    return (size_t)N * S * 2 * k * Co + (size_t)3 * N * k * k * Co + (size_t)N * k * k * Co + (size_t)N * (k + 1) * (k + 1) * Co;
}

hipError_t launch_gen_l0_coord(hipStream_t st, const float* wt0, const float* bias, const float* lin, int L, int S, int Co, int k, float* cterm)
{
    if (k > KMAX || k % 2 == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gen_l0_coord_kernel, dim3(nblk((size_t)S * S * Co)), dim3(256), 0, st, wt0, bias, lin, L, S, Co, k, cterm);
    return hipGetLastError();
}

// scratch: gen_l0_scratch_floats(N, S, Co, k) floats; the forward uses its last two regions (U, PS)
hipError_t launch_gen_l0_fwd(hipStream_t st, const float* z, const float* wt0, const float* cterm, float* scratch, float* out, int N, int L, int S,
                             int Co, int k)
{
    if (k > KMAX || k % 2 == 0) return hipErrorInvalidValue;
    float* u = scratch + (size_t)N * S * 2 * k * Co + (size_t)3 * N * k * k * Co;
    float* ps = u + (size_t)N * k * k * Co;
    hipLaunchKernelGGL(gen_l0_u_kernel, dim3(nblk((size_t)N * k * k * Co)), dim3(256), 0, st, z, wt0, N, L, Co, k, u);
    hipLaunchKernelGGL(gen_l0_prefix_kernel, dim3(nblk((size_t)N * Co)), dim3(256), 0, st, u, N, Co, k, ps);
    if (Co % 4 == 0) {
        const size_t total = (size_t)N * S * S * (Co / 4);
        hipLaunchKernelGGL((gen_l0_fwd_kernel<4>), dim3(nblk(total)), dim3(256), 0, st, ps, cterm, S, Co, k, total, out);
    } else {
        const size_t total = (size_t)N * S * S * Co;
        hipLaunchKernelGGL((gen_l0_fwd_kernel<1>), dim3(nblk(total)), dim3(256), 0, st, ps, cterm, S, Co, k, total, out);
    }
    return hipGetLastError();
}

// dpre: gradient wrt the pre-activation of layer 0 [N][P][Co]; dz (row stride ld) always; weight / bias gradient when alpha != 0
hipError_t launch_gen_l0_bwd(hipStream_t st, const float* dpre, const float* z, const float* wt0, const float* lin, float* scratch, int N, int L,
                             int S, int Co, int k, float alpha, float* gw, float* gb, float* dz, int ld)
{
    if (k > KMAX || k % 2 == 0) return hipErrorInvalidValue;
    float* rows = scratch;
    float* T = scratch + (size_t)N * S * 2 * k * Co;
    hipLaunchKernelGGL(gen_l0_rows_kernel, dim3((unsigned)(N * S)), dim3(256), sizeof(float) * (size_t)(S + 2 * (k / 2)), st, dpre, lin, S, Co, k,
                       rows);
    hipLaunchKernelGGL(gen_l0_taps_kernel, dim3(nblk((size_t)N * k * k * Co)), dim3(256), 0, st, rows, lin, N, S, Co, k, T);
    if (alpha != 0.f)
        hipLaunchKernelGGL(gen_l0_wgrad_kernel, dim3(nblk((size_t)k * k * (L + 2) * Co + Co)), dim3(256), 0, st, T, z, N, L, Co, k, alpha, gw, gb);
    hipLaunchKernelGGL(gen_l0_dz_kernel, dim3((unsigned)(N * L)), dim3(64), 0, st, T, wt0, L, Co, k, ld, dz);
    return hipGetLastError();
}
