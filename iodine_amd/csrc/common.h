// Shared device helpers and launcher declarations for libiodine_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define IOD_DEVINL __device__ __forceinline__

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to (function, device): every launcher instance keeps a bit mask of
// the devices it has configured (`devs`, a function-local static) instead of one process-wide flag, so a second device in the
// same process - tests, one handle per device - gets the attribute too.  Idempotent, so a race between two threads is benign.
inline hipError_t iod_set_max_lds(const void* fn, int bytes, std::atomic<unsigned>& devs)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned bit = 1u << (dev & 31);
    if (devs.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) devs.fetch_or(bit, std::memory_order_release);
    return e;
}
// compute units of the CURRENT device (persistent grids are sized from it), cached per device
inline hipError_t iod_cu_count(int* n_cu)
{
    static std::atomic<int> cache[32];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int v = cache[dev & 31].load(std::memory_order_acquire);
    if (!v) {
        e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        cache[dev & 31].store(v, std::memory_order_release);
    }
    *n_cu = v;
    return hipSuccess;
}

// Timing-only ablation hook for tools/helper_cost.py: a library built with -DIODINE_XSKIP_HOOK lets
// iodine_set_option("xskip", mask) turn the tagged launchers into no-ops (results are then WRONG; the product build has no
// such option).  Bits: 1 partial-tile reductions, 2 head-backward GEMMs, 4 refine head, 8 pixel passes, 16 broadcast layer
// forward, 32 broadcast layer backward, 64 pointwise/axpy, 128 output conv forward, 256 output conv data gradient,
// 512 refinement convs forward, 1024 refinement conv gradients, 2048 output conv weight gradient.
#ifdef IODINE_XSKIP_HOOK
extern int g_iod_xskip;
#define IOD_XSKIP(bit) do { if (g_iod_xskip & (bit)) return hipSuccess; } while (0)
#else
#define IOD_XSKIP(bit) do { } while (0)
#endif

// ELU with alpha = 1 (torch.nn.functional.elu): x > 0 ? x : expm1(x)
IOD_DEVINL float elu1(float v) { return v > 0.f ? v : expm1f(v); }
// the same through v_exp_f32 (absolute error ~1e-7, as in the conv epilogues): the streaming kernels are otherwise VALU-bound on expm1f
IOD_DEVINL float elu1_fast(float v) { return v > 0.f ? v : __expf(v) - 1.f; }
// derivative of ELU expressed through its OUTPUT a = ELU(x): a > 0 ? 1 : a + 1
IOD_DEVINL float elu1_grad_from_out(float a) { return a > 0.f ? 1.f : a + 1.f; }
IOD_DEVINL float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// max over the 64 lanes of a wave, returned to every lane.  DPP steps inside the 16-lane rows, row broadcasts across
// them and one v_readlane - about a dozen VALU instructions; the __shfl_xor butterfly it replaces is six dependent
// ds_bpermute round trips (~100 cycles each), which sat on the staging path of every tile chunk.
IOD_DEVINL float wave_max_f32(float v)
{
    auto dpp = [](float x, auto ctrl, auto rmask) {
        const int r = __builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value,
                                                  decltype(rmask)::value, 0xf, false);
        return __int_as_float(r);
    };
    using std::integral_constant;
    v = fmaxf(v, dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{}));     // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{}));     // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{}));    // row_half_mirror
    v = fmaxf(v, dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{}));    // row_mirror: row max in all 16 lanes
    v = fmaxf(v, dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}));    // row_bcast:15 into rows 1, 3
    v = fmaxf(v, dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}));    // row_bcast:31 into rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Phase timing of the tile / weight-gradient kernels (tools/tile_phase_prof.md; build with IODINE_EXTRA_HIPCC_FLAGS=-DIODINE_TILE_PROF): thread 0 of every
// block accumulates the s_memtime deltas between phase boundaries and writes them to a per-kernel __device__ buffer
// [block][8]; compiled out of the product library.
#ifdef IODINE_TILE_PROF
constexpr int TP_MAXBLK = 16384;
#define TP_DECL unsigned long long tp_last = __builtin_amdgcn_s_memtime(); unsigned tp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TP_STAMP(slot)                                                                      \
    do {                                                                                    \
        const unsigned long long tp_now = __builtin_amdgcn_s_memtime();                     \
        tp_acc[slot] += (unsigned)(tp_now - tp_last);                                       \
        tp_last = tp_now;                                                                   \
    } while (0)
#define TP_FLUSH(buf)                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < TP_MAXBLK)                                         \
        for (int i_ = 0; i_ < 8; ++i_) buf[blockIdx.x * 8 + i_] = tp_acc[i_]
#else
#define TP_DECL
#define TP_STAMP(slot)
#define TP_FLUSH(buf)
#endif


// ---- helpers of the split-fp16 weight-gradient kernels (transposing stagers) ----------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// out[e] = in[(e + r) & 3] with conditional moves only (a runtime-indexed vector would be demoted to scratch)
IOD_DEVINL float4 rot4(const float4 v, int r)
{
    const bool b0 = r & 1, b1 = r & 2;
    float4 t;
    t.x = b0 ? v.y : v.x; t.y = b0 ? v.z : v.y; t.z = b0 ? v.w : v.z; t.w = b0 ? v.x : v.w;
    float4 o;
    o.x = b1 ? t.z : t.x; o.y = b1 ? t.w : t.y; o.z = b1 ? t.x : t.z; o.w = b1 ? t.y : t.w;
    return o;
}

// (x0, x1) -> packed fp16 pairs hi, lo with x = hi + lo: hi = x rounded toward zero to fp16 (v_cvt_pkrtz: the 11 leading bits for the scaled
// range), lo = x - hi (exact in fp32; hipcc emits v_fma_mix_f32 with the fp16 hi as a source and folds the caller's power-of-two scale multiply
// into it), each pair packed by one v_cvt_pkrtz: 6 VALU instructions per pair where the mask-and-subtract form took 8 - every split site was
// instruction-issue-bound (one VALU instruction per 4 cycles and wave).  Same bits as the mask form except below the fp16 normal range, where
// lo now also carries what the conversion of hi dropped.
IOD_DEVINL unsigned pack_hi_lo(float x0, float x1, unsigned& lo_out)
{
    typedef __fp16 h2_ __attribute__((ext_vector_type(2)));
    const h2_ hi = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const h2_ lo = __builtin_amdgcn_cvt_pkrtz(x0 - (float)hi.x, x1 - (float)hi.y);
    unsigned uh;
    __builtin_memcpy(&uh, &hi, 4); __builtin_memcpy(&lo_out, &lo, 4);
    return uh;
}

// power-of-two scale for a tile with max |x| = mx, keeping the current one while mx*cur stays in [2^9, 2^14.5)
IOD_DEVINL float tile_scale(float mx, float cur)
{
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return cur;
    const float t = mx * cur;
    if (t >= 512.f && t < 23170.f) return cur;
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
    int se = 12 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return __uint_as_float((unsigned)(127 + se) << 23);
}

// ---- packed conv-weight geometry (shared by host packer and kernels) ------------------
// A 3x3 conv with CIN (padded) input channels is cut into chunks of CC channels.  Inside a
// chunk the reduction index is organised in "quads" q = tap * (CC/4) + cig: 4 consecutive
// input channels (cig*4 .. +3) at filter tap `tap` (= dy*3+dx).  Quads are consumed in
// pairs by v_mfma_f32_32x32x2_f32: lanes 0-31 feed quad 2g, lanes 32-63 quad 2g+1, element
// s of the quad at MFMA step s.  wpk[chunk][q][co] is a float4 over the quad's 4 channels.
__host__ __device__ constexpr int conv_cc(int cin) { return cin == 20 ? 20 : (cin >= 16 ? 16 : cin); }
__host__ __device__ constexpr int conv_nq(int cin) { return 9 * (conv_cc(cin) / 4); }
__host__ __device__ constexpr int conv_nqp(int cin) { return (conv_nq(cin) + 1) & ~1; }
__host__ __device__ constexpr int conv_nchunk(int cin) { return cin / conv_cc(cin); }
// float4 elements in a packed weight tensor
__host__ __device__ constexpr size_t conv_wpk_elems(int cin, int cout) {
    return (size_t)conv_nchunk(cin) * conv_nqp(cin) * cout;
}

// EPI_L0ROWS: data-gradient form whose result is not stored but reduced to per-row left / interior / right sums
// (rows_p[n][y][tile x][3][C]): the input of the broadcast layer's backward when nothing else needs d(pre-activation 0)
// EPI_L0ROWSX (training): the same plus a fourth sum per row and tile, sum_x lin[x] * value over ALL the tile's columns
// (rows_p[n][y][tile x][4][C]): with it the coordinate-channel gradients of the broadcast layer follow from row sums too
enum ConvEpilogue { EPI_BIAS_ELU = 0, EPI_MUL_ELUGRAD = 1, EPI_NONE = 2, EPI_OUT4 = 3, EPI_L0ROWS = 4, EPI_L0ROWSX = 5 };

// ---- launchers (each returns hipError_t from the launch) -------------------------------
hipError_t launch_pack_conv_weights(hipStream_t st, const float* src_oihw, int O, int I, int cin_pad,
                                    int cout, int transpose_flip, float* dst);
hipError_t launch_conv3x3_tile(hipStream_t st, const float* in, const float* wpk, const float* bias,
                               const float* aux, float* out, int N, int S, int cin, int cout, int epi);
hipError_t launch_conv3x3_gather(hipStream_t st, const float* in, const float* wpk, const float* bias,
                                 float* out, int N, int IH, int IW, int cin, int cout, int stride);
hipError_t launch_dec_out(hipStream_t st, const float* in, const float* wk, const float* bias, float* out,
                          int N, int S, int C);

// kernels_pixel.hip
int pixel_blocks_per_image(int P);
// strict = 1 (conv_precision 0): libm expf + IEEE division in the per-pixel mixture terms (pixel_terms.h), 0: v_exp_f32 / v_rcp_f32 where bounded
hipError_t launch_pixel_pass1(hipStream_t st, const float* x4, const float* dec, float* g, double* part, int B,
                              int K, int P, float sigma, int strict = 0);
hipError_t launch_pixel_finalize(hipStream_t st, const double* part, int B, int K, int P, int use_ln,
                                 float* lnstat, float* ll_img);
// finalize + KL + batch means in one launch; counter: one zero-initialised device word per handle
hipError_t launch_pixel_finalize_elbo(hipStream_t st, const double* part, int B, int K, int P, int use_ln, float* lnstat, float* ll_img,
                                      const float* pm, const float* plv, int L, float* img_terms, float* scal, unsigned* counter);
hipError_t launch_pixel_pass2(hipStream_t st, const float* x4, const float* dec, const float* lnstat,
                              const float* lin, float* enc, int B, int K, int S, float sigma, float* enc_sh = nullptr, unsigned chmask = 0x1ffffu,
                              int strict = 0);
hipError_t launch_final_out(hipStream_t st, const float* dec, float* pred, float* mask, float* mean, float* logits,
                            int B, int K, int P);
// kernels_misc.hip
hipError_t launch_x_to_nhwc4(hipStream_t st, const float* x, float* x4, int B, int P);
hipError_t launch_posterior_init(hipStream_t st, const float* im, const float* ilv, float* pm, float* plv, float* h,
                                 float* c, int N, int L, int H);
hipError_t launch_dec_l0_prepare(hipStream_t st, const float* w, const float* bias, const float* lin, int C, int L,
                                 int S, float* wcls, float* wclsT, float* cmap);
hipError_t launch_dec_v(hipStream_t st, const float* pm, const float* plv, const float* eps, const float* z_in,
                        const float* wcls, float* z_out, float* V, int N, int L, int C);
hipError_t launch_dec_l0(hipStream_t st, const float* V, const float* cmap, float* out, int N, int S, int C, float* tmax = nullptr);
// slot groups of the fused layer-0 reduction (each at most 32 slots); Dpart holds l0_dgroups(N) * P * C floats
inline int l0_dgroups(int N) { const int g = (N + 31) / 32; return g < 8 ? 8 : g; }
// several small device-to-device copies in ONE launch (iodine_set_params: biases and raw weight copies)
// (round 5: also the transposes [R][Cc] -> [Cc][R] (op 1, n = R * Cc, cols = Cc) and the sum of two vectors (op 2, src + src2) of the same
// parameter update - one launch instead of seven)
constexpr int MCOPY_MAX = 40;
struct MultiCopy { const float* src[MCOPY_MAX]; const float* src2[MCOPY_MAX]; float* dst[MCOPY_MAX]; int n[MCOPY_MAX]; int op[MCOPY_MAX]; int cols[MCOPY_MAX]; int count; };
hipError_t launch_multi_copy(hipStream_t st, const MultiCopy& mc);
size_t l0_rows_scratch_floats(int N, int C);
hipError_t launch_l0_reduce_cls_tiles(hipStream_t st, const float* rows_p, float* Rc, int N, int S, int C, float* scratch);
hipError_t launch_l0_reduce_cls_tiles_x(hipStream_t st, const float* rows_p, float* Rc, float* rown, int N, int S, int C, float* scratch);
hipError_t launch_l0_rowsum_acc(hipStream_t st, const float* rown, int N, int S, int C, float alpha, int first, float* Rsum);
hipError_t launch_l0_coord_grads_rows(hipStream_t st, const float* Rsum, const float* lin, int S, int C, int L, float alpha,
                                      float* gw, float* gb);
hipError_t launch_l0_reduce(hipStream_t st, const float* dpre, float* rows, float* Rc, int N, int S, int C, float* Dpart,
                            float* Dacc, float alpha, int first);
hipError_t launch_dz_latent(hipStream_t st, const float* Rc, const float* wclsT, const float* pm, const float* plv,
                            const float* eps, int N, int L, int C, int use_ln, float* g_pm, float* g_plv, float* latent, int Lreal = 0);
hipError_t launch_elbo(hipStream_t st, const float* pm, const float* plv, const float* ll_img, int B, int K, int L,
                       float* img_terms, float* scal);
hipError_t launch_refine_head(hipStream_t st, const float* feat, int N, int PL, int C, int H, int L,
                              const float* mlp_wT, const float* mlp_b, const float* wihT, const float* whhT,
                              const float* lstm_b, const float* wmT, const float* bm, const float* wvT, const float* bv,
                              const float* latent, const float* h_prev, const float* c_prev, float* h_out, float* c_out,
                              float* pm, float* plv, float* sv_pooled, float* sv_s, float* sv_gates, float* sv_xin,
                              float* d_mean, float* d_logvar, float* xh = nullptr, float* gp = nullptr);
hipError_t launch_transpose(hipStream_t st, const float* src, float* dst, int R, int Cc);
hipError_t launch_add2(hipStream_t st, const float* a, const float* b, float* o, int n);
hipError_t launch_pack_dec_out(hipStream_t st, const float* w, float* wk, int C);
hipError_t launch_conv3x3_gather_dgrad(hipStream_t st, const float* d, const float* wpk, const float* aux, float* out,
                                       int N, int big_h, int big_w, int c, int stride);
// kernels_train.hip
int wgrad_tile_blocks(int N, int S);
hipError_t launch_conv3x3_wgrad_tile(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N,
                                     int S, int ci, int nco, int* nparts, int* ncop, int* nbias_parts);
hipError_t launch_conv3x3_wgrad_gather(hipStream_t st, const float* in, const float* d, float* part, int N, int IH,
                                       int IW, int cip, int co, int stride, int* nparts, int* cipad);
constexpr int WGRAD_FOLD = 32;    // tiles left after the first reduction stage; `fold` holds WGRAD_FOLD * 9 * ci_pad * co_pad floats
hipError_t launch_wgrad_reduce(hipStream_t st, const float* part, int nparts, int ci_pad, int co_pad, int O_real,
                               int I_real, int I_dst, float alpha, float* dst, float* fold = nullptr,
                               const float* part_b = nullptr, int nb = 0, float* dst_b = nullptr);
hipError_t launch_colsum(hipStream_t st, const float* src, int rows, int cols, int ld, float alpha, float* dst);
hipError_t launch_colsum_tall(hipStream_t st, const float* src, int rows, int cols, float alpha, float* dst, float* tmp,
                              size_t tmp_elems);
hipError_t launch_sgemm(hipStream_t st, int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                        const float* B, int ldb, float beta, float* C, int ldc);
// C = alpha * A^T . B + beta * C, A [K][M], B [K][N], on fp32 MFMA (M, N multiples of 32); mode 1: C through the broadcast layer's weight map
bool sgemm_tn_mfma_ok(int M, int N, int K);
hipError_t launch_sgemm_tn_mfma(hipStream_t st, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                                float beta, float* C, int ldc, int mode, int mode_c, int a_rowmajor = 0);
hipError_t launch_l0_tap_sums(hipStream_t st, const float* Rc, float* RT, int N, int C);
hipError_t launch_l0_scatter_z(hipStream_t st, const float* tmp, int L, int C, float alpha, float* gw);
hipError_t launch_l0_latent_wgrad(hipStream_t st, const float* Rc, const float* z, int N, int L, int C, float alpha, float* gw);
hipError_t launch_l0_coord_grads(hipStream_t st, const float* D, const float* lin, int S, int C, int L, float alpha,
                                 float* gw, float* gb, float* scratch);
hipError_t launch_loss(hipStream_t st, const float* scal, int n, float* loss);
hipError_t launch_scale(hipStream_t st, const float* a, float alpha, float* o, int n);
hipError_t launch_axpy(hipStream_t st, const float* x, float alpha, float* y, int n);
hipError_t launch_axpy_dev(hipStream_t st, const float* x, float alpha, const float* alpha_dev, float* y, int n, int accumulate);
hipError_t launch_mean2(hipStream_t st, const float* a, const float* b, int n, float* out);
hipError_t launch_lstm_bwd_pointwise(hipStream_t st, const float* gates, const float* c0, const float* c1,
                                     const float* dc1_read, const float* dh1, const float* dc1_carry, float* dgates,
                                     float* dc0, int N, int H);
hipError_t launch_mlp_bwd_pointwise(hipStream_t st, const float* du, int ldu, const float* s, float* ds, int N, int H);
hipError_t launch_pool_bwd(hipStream_t st, const float* dpooled, const float* act, float* dpre, int N, int PL, int C);
bool head_bptt_fits(int L, int H, int Cr);
hipError_t launch_head_bptt(hipStream_t st, const float* g_pm, const float* g_plv, const float* gates, const float* cst, const float* u,
                            const float* Wm, const float* Wv, const float* Whh, const float* Wih, const float* Wmlp, float* ddm,
                            float* ddv, float* dgates, float* ds, float* dpooled, int T, int N, int B, int L, int H, int Cr);
// split-precision (3 x fp16 MFMA) variant of the stride-1 tile conv
hipError_t launch_pack_conv_weights_f16(hipStream_t st, const float* src, int O, int I, int cin, int cout, int tflip,
                                        float* meta, void* dst);
hipError_t launch_conv3x3_tile_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                     const float* bias, const float* aux, float* out, int N, int S, int cin, int cout,
                                     int epi, int rev = 0);
hipError_t launch_adam_multi(hipStream_t st, const long long* ptrs, const long long* offs, int n_tensors, long long total,
                             double lr, double beta1, double beta2, double eps, double wd, int step);
hipError_t launch_randn_philox(hipStream_t st, float* out, long long n, unsigned long long seed, unsigned long long stream_id);
hipError_t launch_ari_table(hipStream_t st, const float* mask, const unsigned char* gt, int B, int K, int G, int P,
                            int* table);
hipError_t launch_pack_dec_out_gemm(hipStream_t st, const float* w, int C, float* meta, void* dst);
hipError_t launch_pack_dec_out_dgrad(hipStream_t st, const float* w, int C, const float* meta, void* dst);
hipError_t launch_dec_out_dgrad_f16x3(hipStream_t st, const float* g, const void* wpk, const float* wmeta, const float* aux,
                                      float* out, int N, int S, int C, float* tmax = nullptr);
hipError_t launch_dec_out_stream_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta,
                                       const float* bias, float* out, int N, int S, int C, const float* tmax = nullptr);

// row-streaming form of the same output conv (round 5: no halo recompute; S in {32, 64, 128}, needs the producer's cell maxima)
bool dec_out_rows_ok(int S, int C, const float* tmax);
// f32 = 1: exact fp32 MFMA form (wpk = launch_pack_dec_out_rows32, wmeta / tmax unused)
hipError_t launch_dec_out_rows_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias, float* out,
                                     int N, int S, int C, const float* tmax, int f32 = 0);
hipError_t launch_pack_dec_out_rows32(hipStream_t st, const float* w, int C, float* dst);

// kernels_convws.hip: weight-stationary split-fp16 3x3 conv C -> C (weights in registers, persistent blocks)
hipError_t launch_pack_conv_weights_ws(hipStream_t st, const float* src, int C, int tflip, float* meta, void* dst);
// kernels_pack.hip: the packs above for many tensors in two launches.  kind 0 = launch_pack_conv_weights_ws(src, C = p[0], tflip = p[1]),
// kind 1 = launch_pack_conv_weights_f16(src, O = p[0], I = p[1], cin = p[2], cout = p[3], tflip = p[4]); meta / dst as there
// round 5: kind 2 = launch_refine_l0_pack(src, O = p[0], cinw = p[1]), kind 3 = launch_pack_dec_out_gemm(src, C = p[0]),
// kind 4 = launch_pack_dec_out_dgrad(src, C = p[0]) with the scale kind 3 of the same batch leaves in the shared meta (p[1] = 1: no scale of its own)
constexpr int PACK_BATCH_MAX = 48;
struct PackJob { const float* src; void* dst; float* meta; int kind; int p[5]; };
hipError_t launch_pack_batch(hipStream_t st, const PackJob* jobs, int n);
// round 6: tensors between the reference's shapes and the padded shapes of the inner handle (DIM_LATENT / MLP_UNITS not multiples of 4)
hipError_t launch_pad_gather(hipStream_t st, const float* src, const int* map, float* dst, int n);
hipError_t launch_pad_scatter(hipStream_t st, const float* src, const int* map, float* dst, int n, int accumulate);
hipError_t launch_resize_rows(hipStream_t st, const float* src, float* dst, long long rows, int w_src, int w_dst);
hipError_t launch_cell_max(hipStream_t st, const float* x, float* tmax, int N, int S, int C);
hipError_t launch_conv3x3_ws_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias,
                                   const float* aux, float* out, const float* tmax_in, float* tmax_out, int N, int S, int c,
                                   int epi, int rev);
// exact-fp32 form of the same kernel (conv_precision 0): fp32 weights in the register layout (same byte count), v_mfma_f32_16x16x4_f32
hipError_t launch_pack_conv_weights_ws32(hipStream_t st, const float* src, int C, int tflip, void* dst);
hipError_t launch_conv3x3_ws_f32(hipStream_t st, const float* in, const void* wpk, const float* bias, const float* aux, float* out,
                                 int N, int S, int c, int epi, int rev);
// kernels_wgrad32.hip: exact-fp32 weight gradient, persistent + prefetched (part: [nparts][9][c][c], part_b: [nbias_parts][c])
hipError_t launch_conv3x3_wgrad_f32_ws(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N, int S, int c,
                                       int* nparts, int* ncop, int* nbias_parts, float alpha = 1.f, int accum = 0);
// exact-fp32 weight gradient of the output conv c -> 4 in GEMM form (part: [nparts][9][c][4], part_b: [nbias_parts][4])
hipError_t launch_dec_out_wgrad_f32(hipStream_t st, const float* a, const float* g, float* part, float* part_b, int N, int S, int c,
                                    int* nparts, int* nbias_parts);
inline size_t conv_ws_wpk_bytes(int C) { return (size_t)(C / 16) * (C / 32) * 9 * 2 * 64 * 16; }
inline size_t conv_ws_tmax_floats(int N, int S) { return (size_t)N * (S / 16) * (S / 8) * 4; }

// tools/experiments/kernels_wino.hip (experiment, only in libraries built by tools/wino_variants.sh): Winograd F(2x2, 3x3) form
size_t conv_wino_wpk_bytes(int C);
size_t conv_wino_scratch_floats(int C);
hipError_t launch_pack_conv_weights_wino(hipStream_t st, const float* src, int C, int tflip, float* meta, void* dst, float* scratch);
hipError_t launch_conv3x3_wino_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias,
                                     const float* aux, float* out, const float* tmax_in, float* tmax_out, int N, int S, int c,
                                     int epi, int rev);

// kernels_refine.hip: split-fp16 stride-2 convs of the refinement network
hipError_t launch_conv3x3_s2_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias,
                                   float* out, int N, int S, int cin_real, int cout, const float* addmap = nullptr, int kdiv = 0,
                                   int f32 = 0);
// exact-fp32 forms of the three stride-2 kernels (conv_precision 0, round 5: v_mfma_f32_32x32x2_f32; f32 = 1 on the launchers):
// fp32 weights in the same LDS-tile layout / byte count as launch_pack_conv_weights_f16
hipError_t launch_pack_conv_weights_s2f32(hipStream_t st, const float* src, int O, int I, int cin, int cout, int tflip, void* dst);
hipError_t launch_ref_split_weights(hipStream_t st, const float* w, int O, float* w_slot, float* w_sh);
hipError_t launch_enc_expand_weights(hipStream_t st, const float* w, int O, int n_in, const int* map17, float* w17, int kk = 9);
hipError_t launch_enc_gather_grad(hipStream_t st, const float* g17, int O, int n_in, const int* map17, float* gw, int kk = 9);

// kernels_generic.hip: fallback fp32 convs for any odd kernel size / channel count (correctness path, see the file header)
constexpr int GEN_WGRAD_SLICES = 64;
constexpr int GEN_WGRAD_OUT_SLICES_MAX = 512;   // row slices of the 4-output-channel GEMM form (one per resident block)
constexpr int GEN_WGRAD_SLICES_MAX = 128;  // the row-staged form sizes its slices to the chip (two blocks per CU): scratch is sized for this many
hipError_t launch_gen_pack_weights(hipStream_t st, const float* w, int Co, int Ci, int k, float* wt);
// chmask (stride-2 MFMA forms only): bit c = input channel c can be non-zero - groups of channels whose weights AND inputs are zero (an
// ARCH.ENCODING subset: absent encoding channels) are skipped; all ones = every channel
hipError_t launch_gen_conv_fwd(hipStream_t st, const float* in, const float* wt, const float* bias, float* out, int N, int Si, int Ci,
                               int ldc, int Co, int k, int s, int elu, unsigned chmask = 0xffffffffu);
hipError_t launch_gen_conv_dgrad(hipStream_t st, const float* dout, const float* wt, const float* aux, float* din, int N, int Si, int Ci,
                                 int ldi, int Co, int k, int s);
size_t gen_wgrad_scratch_floats(int Ci, int Co, int k);
hipError_t launch_gen_conv_wgrad(hipStream_t st, const float* in, const float* dout, float* scratch, int N, int Si, int Ci, int ldc,
                                 int Ci_dst, int Co, int k, int s, float alpha, float* gw, float* gb, unsigned chmask = 0xffffffffu);
// kernels_genl0.hip: the spatial-broadcast layer of the generic decoder without the broadcast tensor (prefix table of per-tap latent products
// forward, tap-window sums of the gradient backward); any odd k <= GEN_L0_KMAX
constexpr int GEN_L0_KMAX = 7;
size_t gen_l0_scratch_floats(int N, int S, int Co, int k);
hipError_t launch_gen_l0_coord(hipStream_t st, const float* wt0, const float* bias, const float* lin, int L, int S, int Co, int k, float* cterm);
hipError_t launch_gen_l0_fwd(hipStream_t st, const float* z, const float* wt0, const float* cterm, float* scratch, float* out, int N, int L, int S,
                             int Co, int k);
hipError_t launch_gen_l0_bwd(hipStream_t st, const float* dpre, const float* z, const float* wt0, const float* lin, float* scratch, int N, int L,
                             int S, int Co, int k, float alpha, float* gw, float* gb, float* dz, int ld);
// kernels_gens2.hip: the stride-2 convs of the generic path on v_mfma_f32_16x16x4_f32 (round 5)
bool gen_s2_fwd_ok(int k, int Ci, int ldc, int Co);
bool gen_s2_dgrad_ok(int k, int Ci, int ldi, int Co);
bool gen_s2_wgrad_ok(int k, int Ci, int ldc, int Co);
hipError_t launch_gen_s2_fwd(hipStream_t st, const float* in, const float* wt, const float* bias, float* out, int N, int Si, int Ci, int ldc,
                             int Co, int k, int elu, unsigned chmask = 0xffffffffu);
hipError_t launch_gen_s2_dgrad(hipStream_t st, const float* dout, const float* wt, const float* aux, float* din, int N, int Si, int Ci, int ldi,
                               int Co, int k);
hipError_t launch_gen_s2_wgrad(hipStream_t st, const float* in, const float* dout, float* part, int N, int Si, int Ci, int ldc, int Co, int k,
                               int nsl_max, int* nsl_out, unsigned chmask = 0xffffffffu);
hipError_t launch_gen_identity(hipStream_t st, float* m, int rows, int L);
hipError_t launch_ref_unsplit_grad(hipStream_t st, const float* g20, int O, float* gw);
hipError_t launch_enc_join(hipStream_t st, const float* enck, const float* encs, float* enc, int N, int K, int P);
hipError_t launch_conv3x3_s2_dgrad_f16x3(hipStream_t st, const float* d, const void* wpk, const float* wmeta, const float* aux,
                                         float* out, int N, int S, int c, int f32 = 0);
hipError_t launch_conv3x3_s2_wgrad_f16x3(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N,
                                         int S, int ci_real, int nco, int* nparts, int* cipad, int* nbias_parts,
                                         const float* a2 = nullptr, int kdiv = 0, int f32 = 0);
// kernels_refl0.hip: encoding (pixel_pass2's channels) + first refinement layer (conv k3 s2 17 -> 64, ELU) for all slots of an image
size_t refine_l0_wpk_bytes(int O);
hipError_t launch_refine_l0_pack(hipStream_t st, const float* w, int O, int cinw, float* meta, void* dst);
bool refine_l0_fused_ok(int S, int c, int K);
hipError_t launch_refine_l0_fused(hipStream_t st, const float* x4, const float* dec, const float* lnstat, const float* lin, const void* wk,
                                  const float* wkmeta, const void* ws, const float* wsmeta, const float* bias, float* out, float* enck,
                                  float* encs, int B, int K, int S, int c, float sigma, unsigned chmask);
// kernels_refws.hip: weight-stationary stride-2 conv C -> C + bias + ELU (refinement layers 1 ..), wpk = launch_pack_conv_weights_ws(w, C, 0)
bool conv3x3_s2ws_ok(int S, int c);
hipError_t launch_conv3x3_s2ws_f16x3(hipStream_t st, const float* in, const void* wpk, const float* wmeta, const float* bias, float* out,
                                     int N, int S, int c, int f32 = 0);       // f32 = 1: exact fp32 form, wpk = launch_pack_conv_weights_ws32(w, c, 0)
// kernels_refbwd.hip: data gradient of refinement layer 1 + weight / bias gradient of layer 0 in one pass (dpre0 never stored)
bool refine_bwd01_ok(int S, int c);
hipError_t launch_refine_bwd01(hipStream_t st, const float* rd1, const void* wpk, const float* wmeta, const float* act0, const float* enck,
                               const float* encs, float* part, float* part_b, int NT, int S, int c, int kdiv, int* nparts, int* cipad,
                               int* nbias_parts);
hipError_t launch_dec_out_wgrad_gemm_f16x3(hipStream_t st, const float* a, const float* g, float* part, float* part_b, int N,
                                           int S, int c, int* nparts, int* nbias_parts);
hipError_t launch_dec_out_bwd_fused_f16x3(hipStream_t st, const float* a, const float* g, const void* wpk, const float* wmeta,
                                          float* out, float* tmax, float* part, float* part_b, int N, int S, int c,
                                          int* nparts, int* nbias_parts, float alpha = 1.f, int accum = 0);
// alpha / accum (round 5): part / part_b = (accum ? part : 0) + alpha x this launch - a block's partial tile kept over the decoder passes
hipError_t launch_conv3x3_wgrad_f16x3_ws(hipStream_t st, const float* a, const float* d, float* part, float* part_b, int N,
                                         int S, int ci, int nco, int* nparts, int* ncop, int* nbias_parts, float alpha = 1.f, int accum = 0);
