#include <hip/hip_runtime.h>
typedef int i32x4_ __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* dst, int n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    i32x4_ r;
    const unsigned long long p = (unsigned long long)src;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
    r.z = n * 4; r.w = 0x00020000;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_base + wv * 1024);
    const unsigned voff = threadIdx.x * 16;
    const int soff = 0;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(voff), "s"(r), "s"(soff) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float4 v = reinterpret_cast<const float4*>(smem)[threadIdx.x];
    reinterpret_cast<float4*>(dst)[threadIdx.x] = v;
    (void)lane;
}
int main()
{
    float *s, *d; const int n = 1024;
    hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
    float h[1024]; for (int i = 0; i < n; ++i) h[i] = i;
    hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d, n);
    float o[1024]; hipMemcpy(o, d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; ++i) bad += o[i] != h[i];
    printf("bad %d  o[0..3] %g %g %g %g  o[260] %g\n", bad, o[0], o[1], o[2], o[3], o[260]);
    return 0;
}
