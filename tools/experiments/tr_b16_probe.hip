// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds halfword h = its own index; lane l passes byte address base + l*8
// (its own 4 consecutive halfwords 4l..4l+3).  Prints the 4 halfwords every lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out, int stride_bytes)
{
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds + threadIdx.x * stride_bytes;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main()
{
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 32}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned short h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("stride %d bytes per lane\n", stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
// build + run:  hipcc --offload-arch=gfx950 -O2 -o tools/experiments/tr_b16_probe tools/experiments/tr_b16_probe.hip && ./tools/experiments/tr_b16_probe
// measured (MI355X): with contiguous 8-byte chunks per lane, lane l of 16-lane group g receives halfwords
//   64 g + (l & 15) + 16 j, j = 0..3  - i.e. column (l & 15) of the group's 4x16 row-major matrix; in general lane i
//   supplies row i / 4, columns 4 (i % 4) .. +3 (its own 8-byte chunk, anywhere in LDS) and lane c receives column c.
