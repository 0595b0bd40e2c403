// Round 6 probe, cfg2 shapes (multi-dSprites 64x64, 32 channels, N = 192 slot-images): is a launch structure with TWO queues worth
// building?  Three questions, each answered as wall time of the same launches on one stream vs on two:
//   (1) weight gradients off the critical path: chain A = 4 data-gradient convs (dependent), chain B = 4 weight gradients + their
//       reductions; serial on one stream vs A on stream a, B on stream b (what a side stream for the wgrad chain of a decoder pass does);
//   (2) two half-batch pipelines: 8 dependent forward convs over N = 192 on one stream vs 8 + 8 over N = 96 on two streams;
//   (3) a latency-bound helper chain (12 launches of the 4-channel output conv's forward at N / 4: small grids) beside the conv chain.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iiodine_amd/csrc tools/experiments/overlap_probe_ds.cpp \
//         -Liodine_amd -liodine_hip -Wl,-rpath,'$ORIGIN/../../iodine_amd' -o tools/experiments/overlap_probe_ds
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static float* dev_rand(size_t n, float lo, float hi, unsigned seed)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * ((s >> 8) * (1.f / 16777216.f)); }
    float* d; CK(hipMalloc((void**)&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 192, S = argc > 2 ? atoi(argv[2]) : 64, C = argc > 3 ? atoi(argv[3]) : 32;
    const size_t big = (size_t)N * S * S * C;
    float* act[3]; float* tm[3];
    for (int i = 0; i < 3; ++i) {
        act[i] = dev_rand(big, -1.f, 1.f, 1 + i);
        CK(hipMalloc((void**)&tm[i], conv_ws_tmax_floats(N, S) * 4));
        CK(launch_cell_max(0, act[i], tm[i], N, S, C));
    }
    float* act2[3]; float* tm2[3];          // second pipeline's buffers (half batch each, but allocated whole)
    for (int i = 0; i < 3; ++i) {
        act2[i] = dev_rand(big, -1.f, 1.f, 11 + i);
        CK(hipMalloc((void**)&tm2[i], conv_ws_tmax_floats(N, S) * 4));
        CK(launch_cell_max(0, act2[i], tm2[i], N, S, C));
    }
    float* w = dev_rand((size_t)C * C * 9, -0.1f, 0.1f, 4);
    float* wo = dev_rand((size_t)4 * C * 9, -0.1f, 0.1f, 5);
    float* bias = dev_rand(C, -0.1f, 0.1f, 6);
    float* o4; CK(hipMalloc((void**)&o4, (size_t)N * S * S * 4 * 4));
    char* wsp; CK(hipMalloc((void**)&wsp, conv_ws_wpk_bytes(C) + 64));
    float* wsmeta = (float*)(wsp + conv_ws_wpk_bytes(C));
    CK(launch_pack_conv_weights_ws(0, w, C, 0, wsmeta, wsp));
    char* op; CK(hipMalloc((void**)&op, 1 << 20)); float* ometa; CK(hipMalloc((void**)&ometa, 64));
    CK(launch_pack_dec_out_gemm(0, wo, C, ometa, op));
    const size_t part_elems = (size_t)1024 * 9 * C * C;
    float *part, *partb, *fold, *gw, *gb;
    CK(hipMalloc((void**)&part, part_elems * 4)); CK(hipMalloc((void**)&partb, 1024 * 64 * 4));
    CK(hipMalloc((void**)&fold, (size_t)WGRAD_FOLD * 9 * C * C * 4));
    CK(hipMalloc((void**)&gw, (size_t)9 * C * C * 4)); CK(hipMalloc((void**)&gb, C * 4));
    CK(hipMemset(gw, 0, (size_t)9 * C * C * 4)); CK(hipMemset(gb, 0, C * 4));
    CK(hipDeviceSynchronize());

    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, ea, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    int np, nc, nb;
    // dependent chains: every launch reads what the previous one wrote
    auto conv_chain = [&](hipStream_t st, float** a, float** t, int n, int len, int epi) {
        for (int i = 0; i < len; ++i)
            CK(launch_conv3x3_ws_f16x3(st, a[i & 1], wsp, wsmeta, epi == 0 ? bias : nullptr, epi == 1 ? a[2] : nullptr, a[(i & 1) ^ 1], t[i & 1], t[(i & 1) ^ 1], n, S, C, epi, i & 1));
    };
    auto wgrad_chain = [&](hipStream_t st, int n, int len) {
        for (int i = 0; i < len; ++i) {
            CK(launch_conv3x3_wgrad_f16x3_ws(st, act[2], act2[2], part, partb, n, S, C, C, &np, &nc, &nb));
            CK(launch_wgrad_reduce(st, part, np, C, nc, C, C, C, 0.5f, gw, fold, partb, nb, gb));
        }
    };
    auto helper_chain = [&](hipStream_t st, int n, int len) {
        for (int i = 0; i < len; ++i) CK(launch_dec_out_rows_f16x3(st, act2[2], op, ometa, bias, o4, n, S, C, tm2[2]));
    };
    auto timeit = [&](const std::function<void()>& enqueue) -> float {
        float sum = 0.f;
        const int reps = 8;
        for (int r = 0; r < reps; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            CK(hipStreamWaitEvent(sb, e0, 0));
            enqueue();
            CK(hipEventRecord(eb, sb));
            CK(hipStreamWaitEvent(sa, eb, 0));
            CK(hipEventRecord(ea, sa));
            CK(hipEventSynchronize(ea));
            float ms; CK(hipEventElapsedTime(&ms, e0, ea));
            if (r >= 3) sum += ms;
        }
        return sum / 5;
    };
    // warm-up (function attributes, clocks)
    for (int i = 0; i < 3; ++i) { conv_chain(sa, act, tm, N, 4, 0); conv_chain(sa, act, tm, N, 4, 1); wgrad_chain(sb, N, 2); helper_chain(sb, N, 2); conv_chain(sb, act2, tm2, N / 2, 2, 0); }
    CK(hipDeviceSynchronize());
    printf("shape: N = %d slot-images, %d x %d, %d channels\n", N, S, S, C);
    const int R = 6;      // decoder passes' worth of launches per measurement
    {
        const float a = timeit([&] { for (int r = 0; r < R; ++r) conv_chain(sa, act, tm, N, 4, 1); });
        const float b = timeit([&] { for (int r = 0; r < R; ++r) wgrad_chain(sa, N, 4); });
        const float ser = timeit([&] { for (int r = 0; r < R; ++r) { conv_chain(sa, act, tm, N, 4, 1); wgrad_chain(sa, N, 4); } });
        const float par = timeit([&] { for (int r = 0; r < R; ++r) { conv_chain(sa, act, tm, N, 4, 1); wgrad_chain(sb, N, 4); } });
        printf("(1) %d x [4 dgrad convs] alone %.3f ms (%.1f us each); %d x [4 wgrad + reduce] alone %.3f ms (%.1f us per pair); one stream %.3f ms; two streams %.3f ms  (gain %.1f %% of the serial time)\n",
               R, a, a * 1e3f / (4 * R), R, b, b * 1e3f / (4 * R), ser, par, 100.f * (ser - par) / ser);
    }
    {
        const float full = timeit([&] { for (int r = 0; r < R; ++r) conv_chain(sa, act, tm, N, 8, 0); });
        const float half1 = timeit([&] { for (int r = 0; r < R; ++r) conv_chain(sa, act, tm, N / 2, 8, 0); });
        const float half2 = timeit([&] { for (int r = 0; r < R; ++r) { conv_chain(sa, act, tm, N / 2, 8, 0); conv_chain(sb, act2, tm2, N / 2, 8, 0); } });
        const float quad = timeit([&] { for (int r = 0; r < R; ++r) { conv_chain(sa, act, tm, N / 2, 4, 0); conv_chain(sb, act2, tm2, N / 2, 4, 0); conv_chain(sa, act, tm, N / 2, 4, 0); conv_chain(sb, act2, tm2, N / 2, 4, 0); } });
        printf("(2) %d x [8 forward convs]: N = %d on one stream %.3f ms (%.1f us each); N = %d on one stream %.3f ms (%.1f us each); two pipelines of N = %d on two streams %.3f ms (gain %.1f %% vs one full-batch stream); interleaved enqueue %.3f ms\n",
               R, N, full, full * 1e3f / (8 * R), N / 2, half1, half1 * 1e3f / (8 * R), N / 2, half2, 100.f * (full - half2) / full, quad);
    }
    {
        const float a = timeit([&] { for (int r = 0; r < R; ++r) conv_chain(sa, act, tm, N, 4, 0); });
        const float b = timeit([&] { for (int r = 0; r < R; ++r) helper_chain(sa, N / 4, 12); });
        const float ser = timeit([&] { for (int r = 0; r < R; ++r) { conv_chain(sa, act, tm, N, 4, 0); helper_chain(sa, N / 4, 12); } });
        const float par = timeit([&] { for (int r = 0; r < R; ++r) { conv_chain(sa, act, tm, N, 4, 0); helper_chain(sb, N / 4, 12); } });
        printf("(3) %d x [4 forward convs] alone %.3f ms; %d x [12 small helper launches] alone %.3f ms (%.1f us each); one stream %.3f ms; two streams %.3f ms (gain %.1f %%)\n",
               R, a, R, b, b * 1e3f / (12 * R), ser, par, 100.f * (ser - par) / ser);
    }
    return 0;
}
