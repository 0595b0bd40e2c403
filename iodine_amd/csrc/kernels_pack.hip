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
