// Do the off-critical-path weight gradients overlap with the memory-bound helper kernels of the decoder pass?
// Runs the warp-specialised 64->64 weight gradient on stream A and one helper kernel (several launches) on stream B, alone
// and together, and prints the wall time of each arrangement.  Links against the built library's launchers:
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iiodine_amd/csrc tools/experiments/overlap_probe.cpp \
//         -Liodine_amd -liodine_hip -Wl,-rpath,'$ORIGIN/../../iodine_amd' -o tools/experiments/overlap_probe
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
    const int N = argc > 1 ? atoi(argv[1]) : 224, S = 128, C = 64;
    const size_t big = (size_t)N * S * S * C;
    float* act = dev_rand(big, -1.f, 1.f, 1);
    float* dpre = dev_rand(big, -1.f, 1.f, 2);
    float* out; CK(hipMalloc((void**)&out, big * 4));
    float* out2; CK(hipMalloc((void**)&out2, big * 4));
    float* g4 = dev_rand((size_t)N * S * S * 4, -1.f, 1.f, 3);
    float* o4; CK(hipMalloc((void**)&o4, (size_t)N * S * S * 4 * 4));
    float* w = dev_rand((size_t)C * C * 9, -0.1f, 0.1f, 4);
    float* wo = dev_rand((size_t)4 * C * 9, -0.1f, 0.1f, 5);
    float* bias = dev_rand(C, -0.1f, 0.1f, 6);
    float* enc = dev_rand((size_t)N * S * S * 20, -1.f, 1.f, 7);
    float* w20 = dev_rand((size_t)C * 17 * 9, -0.1f, 0.1f, 8);
    // packs
    char* wsp; CK(hipMalloc((void**)&wsp, conv_ws_wpk_bytes(C) + 64));
    float* wsmeta = (float*)(wsp + conv_ws_wpk_bytes(C));
    CK(launch_pack_conv_weights_ws(0, w, C, 0, wsmeta, wsp));
    float *tin, *tout; CK(hipMalloc((void**)&tin, conv_ws_tmax_floats(N, S) * 4)); CK(hipMalloc((void**)&tout, conv_ws_tmax_floats(N, S) * 4));
    CK(launch_cell_max(0, act, tin, N, S, C));
    char* op; CK(hipMalloc((void**)&op, 1 << 20)); float* ometa; CK(hipMalloc((void**)&ometa, 64));
    CK(launch_pack_dec_out_gemm(0, wo, C, ometa, op));
    char* odp; CK(hipMalloc((void**)&odp, 1 << 20));
    CK(launch_pack_dec_out_dgrad(0, wo, C, ometa, odp));
    char* s2p; CK(hipMalloc((void**)&s2p, (size_t)2 * 9 * 2 * 2 * C * 16 + 64)); float* s2meta = (float*)(s2p + (size_t)2 * 9 * 2 * 2 * C * 16);
    CK(launch_pack_conv_weights_f16(0, w20, C, 17, 32, C, 0, s2meta, s2p));
    float* s2out; CK(hipMalloc((void**)&s2out, (size_t)N * 64 * 64 * C * 4));
    const size_t part_elems = (size_t)512 * 4 * 9 * C * C;
    float *partA, *partbA, *partB, *partbB;
    CK(hipMalloc((void**)&partA, part_elems * 4)); CK(hipMalloc((void**)&partbA, 512 * 64 * 4));
    CK(hipMalloc((void**)&partB, part_elems * 4)); CK(hipMalloc((void**)&partbB, 512 * 64 * 4));
    float *rows, *Rc; CK(hipMalloc((void**)&rows, (size_t)N * S * 3 * C * 4 * 4)); CK(hipMalloc((void**)&Rc, (size_t)N * 9 * C * 4 * 4));
    CK(hipDeviceSynchronize());

    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, ea, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    int np, nc, nb;
    auto wgrad = [&](hipStream_t st) { CK(launch_conv3x3_wgrad_f16x3_ws(st, act, dpre, partA, partbA, N, S, C, C, &np, &nc, &nb, 2)); };
    struct H { std::string name; std::function<void(hipStream_t)> f; };
    std::vector<H> helpers = {
        {"dec_out_stream", [&](hipStream_t st) { CK(launch_dec_out_stream_f16x3(st, act, op, ometa, bias, o4, N, S, C)); }},
        {"dec_out_dgrad", [&](hipStream_t st) { CK(launch_dec_out_dgrad_f16x3(st, g4, odp, ometa, act, out, N, S, C, tout)); }},
        {"dec_out_wgrad", [&](hipStream_t st) { CK(launch_dec_out_wgrad_gemm_f16x3(st, act, g4, partB, partbB, N, S, C, &np, &nb)); }},
        {"s2conv_20_64", [&](hipStream_t st) { CK(launch_conv3x3_s2_f16x3(st, enc, s2p, s2meta, bias, s2out, N, S, 20, C)); }},
        {"l0_reduce", [&](hipStream_t st) { CK(launch_l0_reduce(st, dpre, rows, Rc, N, S, C, nullptr, nullptr, 0.f, 1)); }},
        {"conv_ws_fwd", [&](hipStream_t st) { CK(launch_conv3x3_ws_f16x3(st, act, wsp, wsmeta, bias, nullptr, out2, tin, tout, N, S, C, 0, 0)); }},
    };
    auto run = [&](int na, const std::function<void(hipStream_t)>* hb, int nbk) -> float {
        float best = 1e30f, sum = 0.f;
        const int reps = 6;
        for (int r = 0; r < reps; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            CK(hipStreamWaitEvent(sb, e0, 0));
            // interleave the enqueue order so neither queue starts far ahead
            int ia = 0, ib = 0;
            while (ia < na || ib < nbk) {
                if (ia < na) { wgrad(sa); ++ia; }
                for (int k = 0; k < (na ? (nbk + na - 1) / na : nbk) && ib < nbk; ++k) { (*hb)(sb); ++ib; }
            }
            CK(hipEventRecord(ea, sa)); CK(hipEventRecord(eb, sb));
            CK(hipStreamWaitEvent(sa, eb, 0));
            CK(hipEventRecord(ea, sa));
            CK(hipEventSynchronize(ea));
            float ms; CK(hipEventElapsedTime(&ms, e0, ea));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        return sum / 4;
    };
    // warm
    for (int i = 0; i < 3; ++i) { wgrad(sa); for (auto& h : helpers) h.f(sb); }
    CK(hipDeviceSynchronize());
    const int NA = 6;
    const float ta = run(NA, nullptr, 0);
    printf("wgrad_ws alone: %d launches %.3f ms (%.3f each)\n", NA, ta, ta / NA);
    for (auto& h : helpers) {
        const float t1 = run(0, &h.f, 4) / 4;
        int k = (int)(ta / t1 + 0.5f); if (k < 1) k = 1;
        const float tb = run(0, &h.f, k);
        const float both = run(NA, &h.f, k);
        printf("%-16s %.3f ms each; alone x%d %.3f ms; wgrad x%d alone %.3f; together %.3f  (sum %.3f, max %.3f, overlap gain %.1f %% of sum)\n",
               h.name.c_str(), t1, k, tb, NA, ta, both, ta + tb, ta > tb ? ta : tb, 100.f * (ta + tb - both) / (ta + tb));
    }
    return 0;
}
