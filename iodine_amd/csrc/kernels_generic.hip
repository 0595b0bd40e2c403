// Generic (any odd kernel size, any channel count) fp32 convolution kernels: the FALLBACK path of the library for ARCH
// combinations the tuned gfx950 kernels are not built for - KERNEL_SIZE 5 / 7 (the reference's default DEC.KERNEL_SIZE is 5,
// lib/config/defaults.py:100; configs/test.yaml:40,44) and CONV_CHAN other than 32 / 64.  Reference call sites: nn.Conv2d + F.elu of
// MultiLayerConv (iodine.py:570-594), the output conv (iodine.py:422,435), SpatialBroadcast (iodine.py:505-540) and their autograd.
//
// These kernels are written for correctness, not speed: one thread per output element, plain fp32 FMAs in a fixed order
// (deterministic), no MFMA, no LDS tiling.  The spatial-broadcast layer is MATERIALISED here ([N][P][L+2]) and convolved like any
// other layer; its gradient wrt z is the pixel sum of the data gradient.  Every shipped / benchmarked configuration stays on the
// tuned path (iodine_api.cpp: `generic` is false for KERNEL_SIZE 3 with 32 / 64 channels).
//
// Layouts: activations NHWC with a channel stride `ldc` >= Ci (the 17-of-20 refinement input); weights re-packed at set_params to
// [tap = ky * k + kx][ci][co] (co fastest: coalesced over the threads of a pixel); stride s in {1, 2}, padding k / 2.
#include "common.h"

namespace {

IOD_DEVINL float gen_elu(float v) { return v > 0.f ? v : expm1f(v); }

// w [Co][Ci][k][k] -> wt [k*k][Ci][Co]
__global__ void gen_pack_weights_kernel(const float* __restrict__ w, int Co, int Ci, int kk, float* __restrict__ wt)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Co * Ci * kk) return;
    const int co = idx % Co, ci = (idx / Co) % Ci, tap = idx / (Co * Ci);
    wt[idx] = w[((size_t)co * Ci + ci) * kk + tap];
}

// bc[n][p][0..L-1] = z[n], bc[n][p][L] = x coordinate, bc[n][p][L+1] = y coordinate (SpatialBroadcast, iodine.py:505-540)
__global__ void gen_broadcast_kernel(const float* __restrict__ z, const float* __restrict__ lin, int L, int S, size_t total,
                                     float* __restrict__ bc)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % (L + 2));
    const size_t px = idx / (L + 2);
    const int p = (int)(px % ((size_t)S * S));
    const size_t n = px / ((size_t)S * S);
    bc[idx] = c < L ? z[n * L + c] : (c == L ? lin[p % S] : lin[p / S]);
}

// out[n][oy][ox][co] = act(bias[co] + sum_{tap, ci} in[n][oy*s + ky - pad][ox*s + kx - pad][ci] * wt[tap][ci][co])
__global__ void gen_conv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                                    float* __restrict__ out, int Si, int So, int Ci, int ldc, int Co, int k, int s, int elu, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % Co);
    size_t r = idx / Co;
    const int ox = (int)(r % So); r /= So;
    const int oy = (int)(r % So);
    const size_t n = r / So;
    const int pad = k / 2;
    float acc = bias ? bias[co] : 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy * s + ky - pad;
        if (iy < 0 || iy >= Si) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox * s + kx - pad;
            if (ix < 0 || ix >= Si) continue;
            const float* ip = in + ((n * Si + iy) * Si + ix) * (size_t)ldc;
            const float* wp = wt + ((size_t)(ky * k + kx) * Ci) * Co + co;
            for (int ci = 0; ci < Ci; ++ci) acc = fmaf(ip[ci], wp[(size_t)ci * Co], acc);
        }
    }
    out[idx] = elu ? gen_elu(acc) : acc;
}

// din[n][y][x][ci] = f'(aux) * sum_{tap, co} dout[n][oy][ox][co] * wt[tap][ci][co]   with oy * s + ky - pad = y (ox likewise);
// aux = the layer's INPUT activation (an ELU output): f' = aux > 0 ? 1 : aux + 1; aux = NULL: no factor.  ldi = channel stride of din
// = input channels of the PACKED weight; Ci <= ldi channels are computed (the broadcast layer needs only its L latent channels).
__global__ void gen_conv_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ wt, const float* __restrict__ aux,
                                      float* __restrict__ din, int Si, int So, int Ci, int ldi, int Co, int k, int s, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ci = (int)(idx % Ci);
    size_t r = idx / Ci;
    const int x = (int)(r % Si); r /= Si;
    const int y = (int)(r % Si);
    const size_t n = r / Si;
    const int pad = k / 2;
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int ty = y + pad - ky;
        if (ty < 0 || ty % s != 0 || ty / s >= So) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int tx = x + pad - kx;
            if (tx < 0 || tx % s != 0 || tx / s >= So) continue;
            const float* dp = dout + ((n * So + ty / s) * So + tx / s) * (size_t)Co;
            const float* wp = wt + ((size_t)(ky * k + kx) * ldi + ci) * Co;
            for (int co = 0; co < Co; ++co) acc = fmaf(dp[co], wp[co], acc);
        }
    }
    const size_t o = ((n * Si + y) * Si + x) * (size_t)ldi + ci;
    if (aux) { const float a = aux[o]; acc *= a > 0.f ? 1.f : a + 1.f; }
    din[o] = acc;
}

// partial[slice][tap][ci][co] = sum over the slice's output pixels (fixed order) of in[...][ci] * dout[...][co];
// pseudo-tap k*k, ci = 0: the bias gradient sum dout[...][co]
__global__ void gen_conv_wgrad_partial_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ part,
                                              int N, int Si, int So, int Ci, int ldc, int Co, int k, int s, int nslice)
{
    const int kk = k * k, per = kk * Ci * Co + Co;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nslice * per) return;
    const int e = (int)(idx % per), slice = (int)(idx / per);
    const size_t npx = (size_t)N * So * So, chunk = (npx + nslice - 1) / nslice;
    const size_t q0 = (size_t)slice * chunk, q1 = q0 + chunk < npx ? q0 + chunk : npx;
    float acc = 0.f;
    if (e >= kk * Ci * Co) {
        const int co = e - kk * Ci * Co;
        for (size_t q = q0; q < q1; ++q) acc += dout[q * Co + co];
    } else {
        const int co = e % Co, ci = (e / Co) % Ci, tap = e / (Co * Ci);
        const int ky = tap / k, kx = tap % k, pad = k / 2;
        for (size_t q = q0; q < q1; ++q) {
            const int ox = (int)(q % So), oy = (int)((q / So) % So);
            const size_t n = q / ((size_t)So * So);
            const int iy = oy * s + ky - pad, ix = ox * s + kx - pad;
            if (iy < 0 || iy >= Si || ix < 0 || ix >= Si) continue;
            acc = fmaf(in[((n * Si + iy) * Si + ix) * (size_t)ldc + ci], dout[q * Co + co], acc);
        }
    }
    part[idx] = acc;
}

// gw[co][ci_dst][tap] += alpha * sum_slice partial (OIHW, ci < Ci_dst kept), gb[co] += alpha * sum_slice partial bias
__global__ void gen_conv_wgrad_reduce_kernel(const float* __restrict__ part, int nslice, int Ci, int Ci_dst, int Co, int kk, float alpha,
                                             float* __restrict__ gw, float* __restrict__ gb)
{
    const int per = kk * Ci * Co + Co;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= per) return;
    float acc = 0.f;
    for (int sl = 0; sl < nslice; ++sl) acc += part[(size_t)sl * per + e];
    if (e >= kk * Ci * Co) { if (gb) gb[e - kk * Ci * Co] += alpha * acc; return; }
    const int co = e % Co, ci = (e / Co) % Ci, tap = e / (Co * Ci);
    if (ci < Ci_dst) gw[((size_t)co * Ci_dst + ci) * kk + tap] += alpha * acc;
}

// out[n][c] = sum_p src[n][p][c] for c < C (fixed order), row stride ld of src, row stride ldo of out
__global__ void gen_sum_pixels_kernel(const float* __restrict__ src, int P, int C, int ld, int ldo, float* __restrict__ out)
{
    const int n = blockIdx.x, c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* sp = src + (size_t)n * P * ld + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int p = 0;
    for (; p + 3 < P; p += 4) { a0 += sp[(size_t)p * ld]; a1 += sp[(size_t)(p + 1) * ld]; a2 += sp[(size_t)(p + 2) * ld]; a3 += sp[(size_t)(p + 3) * ld]; }
    for (; p < P; ++p) a0 += sp[(size_t)p * ld];
    out[(size_t)n * ldo + c] = (a0 + a1) + (a2 + a3);
}

// rows [rows][L] identity-embedded in [rows][L] zero matrix of height `rows`: the "class-sum to latent" matrix of dz_latent when
// the gradient wrt z already sits in the first L entries of every row
__global__ void gen_identity_kernel(float* __restrict__ m, int rows, int L)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * L) return;
    m[idx] = (idx / L) == (idx % L) ? 1.f : 0.f;
}

inline unsigned gen_blocks(size_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace

hipError_t launch_gen_pack_weights(hipStream_t st, const float* w, int Co, int Ci, int k, float* wt)
{
    hipLaunchKernelGGL(gen_pack_weights_kernel, dim3(gen_blocks((size_t)Co * Ci * k * k)), dim3(256), 0, st, w, Co, Ci, k * k, wt);
    return hipGetLastError();
}

hipError_t launch_gen_broadcast(hipStream_t st, const float* z, const float* lin, int N, int L, int S, float* bc)
{
    const size_t total = (size_t)N * S * S * (L + 2);
    hipLaunchKernelGGL(gen_broadcast_kernel, dim3(gen_blocks(total)), dim3(256), 0, st, z, lin, L, S, total, bc);
    return hipGetLastError();
}

hipError_t launch_gen_conv_fwd(hipStream_t st, const float* in, const float* wt, const float* bias, float* out, int N, int Si, int Ci,
                               int ldc, int Co, int k, int s, int elu)
{
    const int So = (Si - 1) / s + 1;
    const size_t total = (size_t)N * So * So * Co;
    hipLaunchKernelGGL(gen_conv_fwd_kernel, dim3(gen_blocks(total)), dim3(256), 0, st, in, wt, bias, out, Si, So, Ci, ldc, Co, k, s, elu, total);
    return hipGetLastError();
}

hipError_t launch_gen_conv_dgrad(hipStream_t st, const float* dout, const float* wt, const float* aux, float* din, int N, int Si, int Ci,
                                 int ldi, int Co, int k, int s)
{
    const int So = (Si - 1) / s + 1;
    const size_t total = (size_t)N * Si * Si * Ci;
    hipLaunchKernelGGL(gen_conv_dgrad_kernel, dim3(gen_blocks(total)), dim3(256), 0, st, dout, wt, aux, din, Si, So, Ci, ldi, Co, k, s, total);
    return hipGetLastError();
}

size_t gen_wgrad_scratch_floats(int Ci, int Co, int k) { return (size_t)GEN_WGRAD_SLICES * ((size_t)k * k * Ci * Co + Co); }

hipError_t launch_gen_conv_wgrad(hipStream_t st, const float* in, const float* dout, float* scratch, int N, int Si, int Ci, int ldc,
                                 int Ci_dst, int Co, int k, int s, float alpha, float* gw, float* gb)
{
    const int So = (Si - 1) / s + 1;
    const size_t per = (size_t)k * k * Ci * Co + Co;
    hipLaunchKernelGGL(gen_conv_wgrad_partial_kernel, dim3(gen_blocks(per * GEN_WGRAD_SLICES)), dim3(256), 0, st, in, dout, scratch, N, Si,
                       So, Ci, ldc, Co, k, s, GEN_WGRAD_SLICES);
    hipLaunchKernelGGL(gen_conv_wgrad_reduce_kernel, dim3(gen_blocks(per)), dim3(256), 0, st, scratch, GEN_WGRAD_SLICES, Ci, Ci_dst, Co,
                       k * k, alpha, gw, gb);
    return hipGetLastError();
}

hipError_t launch_gen_sum_pixels(hipStream_t st, const float* src, int N, int P, int C, int ld, int ldo, float* out)
{
    hipLaunchKernelGGL(gen_sum_pixels_kernel, dim3(N, (C + 63) / 64), dim3(64), 0, st, src, P, C, ld, ldo, out);
    return hipGetLastError();
}

hipError_t launch_gen_identity(hipStream_t st, float* m, int rows, int L)
{
    hipLaunchKernelGGL(gen_identity_kernel, dim3(gen_blocks((size_t)rows * L)), dim3(256), 0, st, m, rows, L);
    return hipGetLastError();
}
