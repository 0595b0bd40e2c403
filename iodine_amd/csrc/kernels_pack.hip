// iodine_set_params, batched: every split-fp16 weight pack of a parameter update (scale + pack per tensor and direction: ~40 launches of
// 1 - 300 blocks, 3 - 6 us each plus a kernel boundary - 0.2 ms of every training step) in TWO launches: one block per tensor for the
// power-of-two scales, then grid (blocks, tensor) for the packs.  Same element functions as the one-tensor kernels (pack_bodies.h).
#include "common.h"
#include "pack_bodies.h"

namespace {

struct PackJobDev {
    const float* src; _Float16* dst; float* meta;
    int kind, p0, p1, p2, p3, p4;                            // kind 0 (ws): p0 = C, p1 = tflip; kind 1 (f16): p0 = O, p1 = I, p2 = cin, p3 = cout, p4 = tflip
    int nscale;                                              // elements of src the scale is taken over
};
struct PackJobsDev { PackJobDev j[PACK_BATCH_MAX]; };

__global__ __launch_bounds__(1024) void pack_batch_scale_kernel(PackJobsDev jobs)
{
    __shared__ float s_red[16];
    const PackJobDev& j = jobs.j[blockIdx.x];
    if (j.nscale > 0) weight_scale_block(j.src, j.nscale, j.meta, s_red);       // (0: the job shares another job's scale)
}

__global__ __launch_bounds__(256) void pack_batch_kernel(PackJobsDev jobs)
{
    const PackJobDev& j = jobs.j[blockIdx.y];
    const float scale = j.meta[0];
    if (j.kind == 0) {
        const size_t total = pack_ws_total(j.p0);
        for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
            j.dst[idx] = pack_ws_element(j.src, j.p0, j.p1, scale, idx);
    } else if (j.kind == 1) {
        const size_t total = pack_f16_total(j.p2, j.p3);
        for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
            j.dst[idx] = pack_f16_element(j.src, j.p0, j.p1, j.p3, j.p4, scale, idx);
    } else if (j.kind == 2) {
        const size_t total = pack_l0_total(j.p0);
        for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
            j.dst[idx] = pack_l0_element(j.src, j.p1, scale, idx);
    } else if (j.kind == 3) {
        const size_t total = pack_out_gemm_total(j.p0);
        for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
            j.dst[idx] = pack_out_gemm_element(j.src, j.p0, scale, idx);
    } else {
        const size_t total = pack_out_dgrad_total(j.p0);
        for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
            j.dst[idx] = pack_out_dgrad_element(j.src, j.p0, scale, idx);
    }
}

}  // namespace

hipError_t launch_pack_batch(hipStream_t st, const PackJob* jobs, int n)
{
    for (int i0 = 0; i0 < n; i0 += PACK_BATCH_MAX) {
        const int m = std::min(PACK_BATCH_MAX, n - i0);
        PackJobsDev d;
        for (int i = 0; i < m; ++i) {
            const PackJob& s = jobs[i0 + i];
            if (s.kind < 0 || s.kind > 4) return hipErrorInvalidValue;
            // elements the power-of-two scale is taken over (kind 4: none - it uses the scale kind 3 leaves in the shared meta)
            const int nscale = s.kind == 0 ? s.p[0] * s.p[0] * 9 : s.kind == 1 ? s.p[0] * s.p[1] * 9 : s.kind == 2 ? s.p[0] * s.p[1] * 9
                               : s.kind == 3 ? 4 * s.p[0] * 9 : 0;
            d.j[i] = PackJobDev{s.src, (_Float16*)s.dst, s.meta, s.kind, s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], nscale};
        }
        hipLaunchKernelGGL(pack_batch_scale_kernel, dim3(m), dim3(1024), 0, st, d);
        hipLaunchKernelGGL(pack_batch_kernel, dim3(72, m), dim3(256), 0, st, d);
    }
    return hipGetLastError();
}

// ---- round 6: DIM_LATENT / REF.MLP_UNITS that are not multiples of 4 (iodine.py:8-32,446-464 accept any) -----------------------------------
// The kernels of the refinement head move weight rows as 16-byte vectors.  The C ABI keeps the reference's shapes at the boundary; the host
// (iodine_api.cpp, "padded inner handle") runs the path at the next multiples of 4 with zero-filled weights / states and moves tensors between
// the two shapes with the element maps below (built once per handle on the host).
//   dst[i] = map[i] >= 0 ? src[map[i]] : 0                                   (parameters: reference shape -> padded shape)
__global__ void pad_gather_kernel(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = map[i];
        dst[i] = j >= 0 ? src[j] : 0.f;
    }
}
//   dst[map[i]] = (accumulate ? dst[map[i]] : 0) + src[i]   for map[i] >= 0    (gradients: padded shape -> reference shape; targets are unique)
__global__ void pad_scatter_kernel(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst, int n, int accumulate)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = map[i];
        if (j >= 0) dst[j] = (accumulate ? dst[j] : 0.f) + src[i];
    }
}
//   rows of length w_src -> rows of length w_dst: the first min(w_src, w_dst) entries copied, the rest of a longer destination row zero
__global__ void resize_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int w_src, int w_dst)
{
    const long long n = rows * w_dst;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / w_dst;
        const int c = (int)(i - r * w_dst);
        dst[i] = c < w_src ? src[r * w_src + c] : 0.f;
    }
}

hipError_t launch_pad_gather(hipStream_t st, const float* src, const int* map, float* dst, int n)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pad_gather_kernel, dim3(std::min((n + 255) / 256, 1024)), dim3(256), 0, st, src, map, dst, n);
    return hipGetLastError();
}
hipError_t launch_pad_scatter(hipStream_t st, const float* src, const int* map, float* dst, int n, int accumulate)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pad_scatter_kernel, dim3(std::min((n + 255) / 256, 1024)), dim3(256), 0, st, src, map, dst, n, accumulate);
    return hipGetLastError();
}
hipError_t launch_resize_rows(hipStream_t st, const float* src, float* dst, long long rows, int w_src, int w_dst)
{
    const long long n = rows * w_dst;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(resize_rows_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, st, src, dst, rows, w_src, w_dst);
    return hipGetLastError();
}
