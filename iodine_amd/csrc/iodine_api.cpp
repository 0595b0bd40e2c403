// Host side of libiodine_hip.so: handle, parameter repacking, workspace planning and the
// T-step refinement loop (kernel launch sequence).  See include/iodine_hip.h for the ABI and the
// reference call sites each entry point replaces.
#include "../../include/iodine_hip.h"
#include "common.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

std::string g_create_error;

struct ParamInfo {
    std::string name;
    int ndim;
    long long dims[4];
    size_t numel() const { size_t n = 1; for (int i = 0; i < ndim; ++i) n *= (size_t)dims[i]; return n; }
};

// bump allocator over a (possibly NULL = size query) base pointer
struct Arena {
    char* base;
    size_t off = 0;
    explicit Arena(void* b) : base((char*)b) {}
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

struct Buffers {               // workspace carve-up for one batch size / mode
    int B = 0, mode = -1;
    size_t bytes = 0;
    float *x4, *V, *dec_out, *g, *lnstat, *ll_img, *img_terms, *scal, *rows, *rows_p, *Rc, *pm, *plv;
    double* part;
    std::vector<float*> act;                   // decoder activations a[0..Dd-1]   (N,P,Cd)
    float *head_xh = nullptr, *head_gp = nullptr;   // refinement head in three launches: LSTM input rows, gate pre-activations
    float* dpre[2];                            // ping-pong gradient wrt pre-activations
    std::vector<float*> tmax_act;              // per-cell max |act[l]| (4 floats per 8 x 16 cell): tile scales of the weight-stationary conv
    float* tmax_dpre[2] = {nullptr, nullptr};  // the same for the two gradient buffers
    // per-iteration buffers: index i (training keeps all T(+1) copies, inference aliases them)
    std::vector<float*> z, g_pm, g_plv, latent, enc, pooled, u, gates, xin, h, c;
    std::vector<float*> enck, encs;             // split refinement input: per-slot [N][P][12], per-image [B][P][8] (alias enc's memory)
    float* rmap = nullptr;                      // split first refinement layer: conv of the per-image channels, [B][S/2][S/2][Cr]
    std::vector<std::vector<float*>> ract;     // [iter][layer] refinement activations
    // training only
    float *wg_part = nullptr, *wg_part_b = nullptr, *wg_fold = nullptr, *Dsum = nullptr, *Dpart = nullptr, *RT = nullptr, *tmp_lz = nullptr;
    // round 5 (option wgrad_accum): per-layer partial weight-gradient tiles kept over the T + 1 decoder passes of a step (one reduction
    // per layer and step); [0] = the output conv, [l] = decoder layer l
    std::vector<float*> wg_acc, wg_acc_b;
    float *rown = nullptr, *Rsum = nullptr;     // training row-sum form of the broadcast layer's backward (EPI_L0ROWSX)
    float* l0scr = nullptr;                     // partial class sums of the row-sum reductions (l0_rows_scratch_floats)
    float *ddm = nullptr, *ddv = nullptr, *dc1 = nullptr, *dgates = nullptr, *dxin = nullptr, *ds = nullptr,
          *dpooled = nullptr;
    float* carry_h[2] = {nullptr, nullptr};
    float* carry_c[2] = {nullptr, nullptr};
    std::vector<float*> rdpre;                 // gradient wrt refinement pre-activations, per layer
    float *gen_scr = nullptr, *gen_l0 = nullptr;                // generic path: wgrad partials; layer-0 scratch (kernels_genl0.hip)
};

}  // namespace

// HIP-event profiler: when enabled every launch of a category is bracketed by two events on the
// launch stream; totals are read back (after a sync) with iodine_profile_read.
struct ProfCat { std::string name; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; unsigned long long seen = 0, seen_win = 0; };

struct GraphEntry { std::vector<uintptr_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; unsigned long long used = 0; };

struct iodine_handle {
    iodine_config cfg;
    std::string err;
    int profile = 0;                            // 0 off, 1 the dominant conv kernels ("conv_tile_*") only, 2 every category
    int profile_stride = 1;                     // level 1: bracket every n-th launch of a category only - an event pair costs ~12 us of idle GPU
                                                // (a barrier packet in front of the kernel and one behind it: tools/step_timeline.py), 54 pairs per cfg3
                                                // training step = 0.8 ms of the step being measured; a stride coprime to the layer count samples every layer
    std::vector<ProfCat> prof;
    ProfCat* prof_cat(const char* name) {
        for (auto& c : prof) if (c.name == name) return &c;
        prof.push_back(ProfCat()); prof.back().name = name; return &prof.back();
    }
    int L, T, K, S, P, Cd, Dd, Cr, Dr, H;
    std::vector<ParamInfo> params;
    bool params_set = false;
    int stop_after = -1;

    // parameter-derived device buffers (owned)
    float* lin = nullptr;                       // linspace(-1,1,S)
    float *wcls = nullptr, *wclsT = nullptr, *cmap = nullptr;
    std::vector<float*> dec_wf, dec_wb, dec_b;  // packed fwd / dgrad weights + bias copies for layers 1..Dd-1
    std::vector<float*> dec_wf16, dec_wb16, dec_wmeta;   // split-fp16 packs (+ {scale, 1/scale, scale_b, 1/scale_b})
    std::vector<float*> dec_wsf, dec_wsb;                // the same weights in the register layout of the weight-stationary conv
    int precision = 1;                          // 0: exact fp32 MFMA, 1: 3 x fp16 MFMA split (fp32-class accuracy)
    int out_bwd_fused = 1;                      // training: output conv data + weight gradient in one pass over the activation
    int fuse_l0 = 1;                            // inference: layer-1 data gradient reduces straight to the layer-0 row sums
    int refine_split = 1;                       // first refinement layer split into a per-slot and a per-image part (split-fp16 path)
    int head_fused = 1;                         // training backward: the head's BPTT recurrence as ONE launch (0 = 9 launches per iteration)
    int refine_bwd_fused = 1;                   // training backward: data gradient of refinement layer 1 + weight gradient of layer 0 in one
                                                // launch, d(pre-activation 0) never stored (kernels_refbwd.hip; 0 = the two launches)
    bool fwd_split = false;                     // the form the saved training forward used
    int variant = 6;                            // split-fp16 stride-1 conv: 6 = weight-stationary persistent kernel (power-of-two image sizes;
                                                // other sizes use 1), 1 = LDS-tiled 16x16 tiles (2 blocks/CU)
    float* dec_out_w32 = nullptr;               // fp32 operand of the row-streaming output conv (conv_precision 0)
    float *dec_out_w = nullptr, *dec_out_b = nullptr, *dec_out_wb = nullptr, *dec_out_w16 = nullptr, *dec_out_meta = nullptr,
          *dec_out_wb16 = nullptr;               // split-fp16 pack of the output conv for its data gradient
    std::vector<float*> ref_w, ref_b;
    float *mlp_wT = nullptr, *mlp_b = nullptr, *wihT = nullptr, *whhT = nullptr, *lstm_b = nullptr;
    float *wmT = nullptr, *bm = nullptr, *wvT = nullptr, *bv = nullptr, *init_mean = nullptr, *init_logvar = nullptr;
    // training: raw copies used by the head backward GEMMs, strided-dgrad packs, gradient accumulators
    float *raw_mlp_w = nullptr, *raw_wih = nullptr, *raw_whh = nullptr, *raw_wm = nullptr, *raw_wv = nullptr;
    std::vector<float*> ref_wb;
    std::vector<float*> ref_wf16, ref_wb16, ref_wmeta;     // split-fp16 packs of the stride-2 convs (+ {scale, 1/scale} x {fwd, dgrad})
    float *ref_wk = nullptr, *ref_wsh = nullptr;           // split first layer: weights in the internal channel order [Cr][12][9], [Cr][8][9]
    float *ref_wk16 = nullptr, *ref_wsh16 = nullptr, *ref_wkmeta = nullptr, *ref_wshmeta = nullptr;   // and their packs
    float* ref_g20 = nullptr;                              // [Cr][20][9] weight gradient in the internal order
    unsigned* elbo_counter = nullptr;                      // ticket of pixel_finalize_elbo_kernel (zero between launches)
    int head_mfma = 1;                                     // LSTM gate pre-activations of the refinement head as one fp32-MFMA GEMM over all slots
    int refine_l0_fused = 1;                               // encoding + first refinement layer in one kernel (kernels_refl0.hip); 0: pixel_pass2 + two convs
    void *ref_l0k = nullptr, *ref_l0s = nullptr; float *ref_l0kmeta = nullptr, *ref_l0smeta = nullptr;     // its weight packs
    int wgrad_accum = 0;                                   // 1: the decoder's partial weight-gradient tiles accumulate over the T + 1 passes of a training
                                                           // step (alpha = pass weight) and are reduced once per layer and step.  Measured (round 5, same
                                                           // process A/B): cfg3 48.197 vs 48.200 ms, cfg2 6.55 vs 6.53 ms - the read-modify-write of the
                                                           // partial tiles in the kernels' tails costs what the 20 saved reduce launches cost: NOT adopted,
                                                           // kept as an option (off: no extra workspace)
    int dec_out_rows = 1;                                  // output conv forward: row-streaming kernel without halo recompute (S in {32, 64, 128}); 0 = 16 x 16 tiles
    int refine_ws = 1;                                     // forward stride-2 convs of refinement layers 1 .. on the weight-stationary kernel (kernels_refws.hip)
    std::vector<float*> ref_wsf, ref_wsf_meta;             // their weights in its register layout
    float *ref_w1ws = nullptr, *ref_w1ws_meta = nullptr;   // layer 1's weights in the register layout of the fused layer-1/0 backward
    // ARCH.ENCODING subsets: reference input channel j of the first refinement layer = internal channel enc_map[j] (code order of
    // iodine.py:277-340); n_in < 17 -> weights expanded to / gradients gathered from the 17 internal channels
    int n_in = 17;
    int enc_map[17];
    unsigned enc_chmask = 0x1ffffu;                        // bit c: internal channel c is part of the encoding (absent ones are written as 0)
    float *ref_w17 = nullptr, *ref_g17 = nullptr;          // [Cr][17][kr * kr]
    // GENERIC fallback path (kernels_generic.hip): KERNEL_SIZE other than 3 or CONV_CHAN other than 32 / 64.  Weights re-packed to
    // [tap][ci][co]; the broadcast layer is materialised; nothing of the tuned conv kernels runs.
    bool generic = false;                                  // the DECODER runs on the generic path
    bool gen_ref = false;                                  // the REFINEMENT conv stack runs on the generic path (round 5: decided separately -
                                                           // the reference's default ARCH has REF.KERNEL_SIZE 3 / 32 channels beside DEC.KERNEL_SIZE 5)
    int kd = 3, kr = 3;                                    // DEC / REF kernel sizes
    int rs = 2;                                            // REF.STRIDE (round 6: other strides run on the generic path's kernels)
    std::vector<float*> gen_wdec, gen_wref;                // [layer]: packed weights
    float *gen_wout = nullptr, *gen_cterm = nullptr, *gen_ident = nullptr;   // output conv pack, [P][Cd] bias + coordinate term of decoder layer 0, [9 Cd][L] identity
    std::vector<float*> gacc;                   // one per parameter, reference shapes (slices of gacc_arena)
    float* gacc_arena = nullptr;
    size_t gacc_total = 0;
    bool fwd_done = false;
    int fwd_batch = 0;
    // last elbo() call (iodine.py:161-241): which z buffer / batch the decoder output in buf.dec_out belongs to
    int last_elbo_iter = -1, last_elbo_batch = 0;
    bool enc_valid = false;                     // the last call left the refinement input ("enc") of its iterations in the workspace
    // hipGraph replay of the fixed-shape launch sequences (option "graph"): one instantiated graph per distinct argument tuple
    int graph = 0;
    std::vector<GraphEntry> graphs;
    std::vector<std::vector<uintptr_t>> seen_keys;       // argument tuples that ran eagerly once (the next call captures)
    unsigned long long graph_clock = 0;
    long long graph_replays = 0, graph_captures = 0;
    std::vector<void*> owned;
    // round 6: DIM_LATENT / REF.MLP_UNITS that are not multiples of 4.  Lreal: the reference's DIM_LATENT when this handle runs at a padded
    // latent width (the 3-D layer-norm and the logger means are taken over the real entries); shim: this handle is only the boundary of a
    // padded INNER handle (see PadShim below)
    int Lreal = 0;
    struct PadShim* shim = nullptr;

    // workspace
    void* ws_user = nullptr; size_t ws_user_bytes = 0;
    void* ws_own = nullptr; size_t ws_own_bytes = 0;
    Buffers buf;

    int fail(int code, const std::string& m) { err = m; return code; }
};

#ifdef IODINE_XSKIP_HOOK
int g_iod_xskip = 0;
#endif

namespace {

hipError_t conv_f16x3(const iodine_handle* h, hipStream_t st, const float* in, const void* wpk, const void* wpk_ws,
                      const float* wmeta, const float* bias, const float* aux, float* out, const float* tmax_in, float* tmax_out,
                      int N, int S, int cin, int cout, int epi, int layer);

#define HIPCHK(h, expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return (h)->fail(IODINE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// bracket one launch expression with profiler events (no-op unless profiling is on)
#define PROF(h, st, cat, expr)                                                       \
    do {                                                                             \
        hipEvent_t e0_ = nullptr, e1_ = nullptr;                                     \
        ProfCat* pc_ = nullptr;                                                      \
        if ((h)->profile > 1) pc_ = (h)->prof_cat(cat);                              \
        else if ((h)->profile == 1 && !strncmp(cat, "conv_tile_", 10)) {             \
            pc_ = (h)->prof_cat(cat);                                                \
            pc_->seen_win++;                                                         \
            if (pc_->seen++ % (unsigned long long)(h)->profile_stride) pc_ = nullptr;    \
        }                                                                            \
        if (pc_) {                                                                   \
            if (pc_->used == pc_->ev.size()) {                                       \
                hipEvent_t a_, b_;                                                   \
                HIPCHK(h, hipEventCreate(&a_)); HIPCHK(h, hipEventCreate(&b_));      \
                pc_->ev.push_back({a_, b_});                                         \
            }                                                                        \
            e0_ = pc_->ev[pc_->used].first; e1_ = pc_->ev[pc_->used].second;         \
            pc_->used++;                                                             \
            HIPCHK(h, hipEventRecord(e0_, st));                                      \
        }                                                                            \
        HIPCHK(h, expr);                                                             \
        if (e1_) HIPCHK(h, hipEventRecord(e1_, st));                                 \
    } while (0)

bool conv_ws_ok(const iodine_handle* h) { return !h->generic && h->variant == 6 && h->precision == 1 && h->S >= 16 && (h->S & (h->S - 1)) == 0; }
// exact-fp32 path (conv_precision 0): the weight-stationary kernel's fp32 form (v_mfma_f32_16x16x4_f32, no tile scales / side buffers) under the same
// conditions; conv_variant 1 keeps the round-1 LDS-tiled fp32 kernels
bool conv_ws32_ok(const iodine_handle* h) { return !h->generic && h->variant == 6 && h->precision == 0 && h->S >= 16 && (h->S & (h->S - 1)) == 0 && (h->Cd == 64 || h->Cd == 32); }

hipError_t conv_f16x3(const iodine_handle* h, hipStream_t st, const float* in, const void* wpk, const void* wpk_ws,
                      const float* wmeta, const float* bias, const float* aux, float* out, const float* tmax_in, float* tmax_out,
                      int N, int S, int cin, int cout, int epi, int layer)
{
    if (conv_ws_ok(h) && cin == cout)
        return launch_conv3x3_ws_f16x3(st, in, wpk_ws, wmeta, bias, aux, out, tmax_in, tmax_out, N, S, cout, epi, layer & 1);
    // zig-zag: odd decoder layers walk the slot-images backwards (forward pass: l0 writes forwards, layer 1 reads
    // backwards, layer 2 forwards, ...; backward pass the same by layer), so a launch starts on the part of its input
    // that the previous launch wrote last - still in the 256 MiB Infinity Cache - instead of the part written first.
    // Tiles are independent: results do not depend on the order.  Measured -0.3 % on the cfg3 step (same-box A/B).
    return launch_conv3x3_tile_f16x3(st, in, wpk, wmeta, bias, aux, out, N, S, cin, cout, epi, layer & 1);
}

template <typename T>
hipError_t dev_alloc(iodine_handle* h, T** p, size_t n)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) { *p = (T*)q; h->owned.push_back(q); }
    return e;
}

void build_param_table(iodine_handle* h)
{
    // names / shapes of the reference module tree in named_parameters() order
    // (iodine.py:26-33, 412-423, 446-464, 543-557, 570-584, 596-604)
    auto add = [&](const std::string& n, std::initializer_list<long long> d) {
        ParamInfo p; p.name = n; p.ndim = (int)d.size(); int i = 0;
        for (long long v : d) p.dims[i++] = v;
        for (; i < 4; ++i) p.dims[i] = 1;
        h->params.push_back(p);
    };
    const iodine_config& c = h->cfg;
    int cin = h->n_in;
    for (int i = 0; i < c.ref_conv_layers; ++i) {
        add("refine.mlc.layers." + std::to_string(i) + ".weight", {c.ref_conv_chan, cin, c.ref_kernel_size, c.ref_kernel_size});
        add("refine.mlc.layers." + std::to_string(i) + ".bias", {c.ref_conv_chan});
        cin = c.ref_conv_chan;
    }
    const long long H = c.ref_mlp_units, L = c.dim_latent;
    add("refine.mlp.layers.0.weight", {H, c.ref_conv_chan});
    add("refine.mlp.layers.0.bias", {H});
    add("refine.lstm.weight_ih", {4 * H, H + 4 * L});
    add("refine.lstm.weight_hh", {4 * H, H});
    add("refine.lstm.bias_ih", {4 * H});
    add("refine.lstm.bias_hh", {4 * H});
    add("refine.mean_update.weight", {L, H});
    add("refine.mean_update.bias", {L});
    add("refine.logvar_update.weight", {L, H});
    add("refine.logvar_update.bias", {L});
    cin = c.dim_latent + 2;
    for (int i = 0; i < c.dec_conv_layers; ++i) {
        add("decoder.mlc.layers." + std::to_string(i) + ".weight", {c.dec_conv_chan, cin, c.dec_kernel_size, c.dec_kernel_size});
        add("decoder.mlc.layers." + std::to_string(i) + ".bias", {c.dec_conv_chan});
        cin = c.dec_conv_chan;
    }
    add("decoder.conv.weight", {4, c.dec_conv_chan, c.dec_kernel_size, c.dec_kernel_size});
    add("decoder.conv.bias", {4});
    add("posterior.init_mean", {L});
    add("posterior.init_logvar", {L});
}

int param_index(const iodine_handle* h, const std::string& name)
{
    for (size_t i = 0; i < h->params.size(); ++i)
        if (h->params[i].name == name) return (int)i;
    return -1;
}

std::string validate(const iodine_config& c)
{
    char m[256];
    if (c.encoding & ~IODINE_ENC_FULL) return "ARCH.ENCODING has unknown entries";
    if ((c.encoding & (IODINE_ENC_POSTERIOR | IODINE_ENC_GRAD_POST)) != (IODINE_ENC_POSTERIOR | IODINE_ENC_GRAD_POST))
        return "ARCH.ENCODING must contain 'posterior' and 'grad_post' (the LSTM input of the refinement head is built for both; every "
               "shipped config and lib/config/defaults.py:57-80 have them); any subset of the image-shaped entries is accepted";
    if (!(c.encoding & (IODINE_ENC_FULL & ~(IODINE_ENC_POSTERIOR | IODINE_ENC_GRAD_POST))))
        return "ARCH.ENCODING has no image-shaped entry: the refinement conv stack would have no input";
    if (c.img_channels != 3) return "ARCH.IMG_CHANNELS must be 3";
    // KERNEL_SIZE 3 with 32 / 64 channels runs on the tuned kernels; other odd kernel sizes and channel counts on the generic
    // fallback path (kernels_generic.hip)
    for (int k : {c.dec_kernel_size, c.ref_kernel_size})
        if (k != 3 && k != 5 && k != 7) return "KERNEL_SIZE must be 3, 5 or 7 (odd: the reference pads with KERNEL_SIZE // 2, iodine.py:419,580)";
    if (c.ref_stride < 1 || c.ref_stride > 8) return "REF.STRIDE must be in 1..8 (2: tuned kernels; other strides: generic path)";
    if (c.img_size < 8 || c.img_size > 1024) return "ARCH.IMG_SIZE must be in 8..1024 (multiples of 16 run on the tuned kernels, other sizes on the generic path)";
    if (c.dec_conv_chan < 8 || c.dec_conv_chan > 256 || c.dec_conv_chan % 4 != 0) return "DEC.CONV_CHAN must be a multiple of 4 in 8..256";
    if (c.ref_conv_chan < 4 || c.ref_conv_chan > 256 || 256 % c.ref_conv_chan != 0) return "REF.CONV_CHAN must divide 256 (4 ... 256)";
    if (9 * c.dec_conv_chan < c.dim_latent) return "DIM_LATENT must not exceed 9 * DEC.CONV_CHAN";
    if (c.dec_conv_layers < 1 || c.dec_conv_layers > 32) return "DEC.CONV_LAYERS must be in 1..32";
    if (c.ref_conv_layers < 1 || c.ref_conv_layers > 16 || (c.ref_stride == 2 && (c.img_size >> c.ref_conv_layers) < 1)) return "REF.CONV_LAYERS out of range for IMG_SIZE";
    if (c.slots < 1 || c.slots > 16) return "ARCH.SLOTS must be in 1..16 (the per-pixel kernels keep every slot of a pixel in registers: instantiated for K <= 16)";
    if (c.iters < 1) return "ARCH.ITERS must be >= 1";
    // (the refinement head reads its weight rows as 16-byte vectors: widths that are not multiples of 4 run on a zero-padded inner handle,
    // PadShim - round 6)
    if (c.dim_latent < 2 || c.dim_latent > 256) return "ARCH.DIM_LATENT must be in 2..256";
    if (c.ref_mlp_units < 1 || c.ref_mlp_units > 1024) return "REF.MLP_UNITS must be in 1..1024";
    if (!(c.sigma > 0)) { snprintf(m, sizeof m, "ARCH.SIGMA must be > 0 (got %g)", c.sigma); return m; }
    return "";
}

// spatial size of refinement layer l's OUTPUT
// (k x k, pad k // 2, stride rs: floor((s + 2 (k // 2) - k) / rs) + 1 = (s - 1) / rs + 1 for odd k)
int ref_out_size(const iodine_handle* h, int s) { return (s - 1) / h->rs + 1; }
// the split-fp16 stride-2 kernels cover the shipped refinement stacks: 32 or 64 channels, even sizes at every layer
bool refine_f16_ok(const iodine_handle* h);

void plan(const iodine_handle* h, int B, int mode, Arena& a, Buffers& b)
{
    const int N = B * h->K, P = h->P, L = h->L, Cd = h->Cd, Cr = h->Cr, H = h->H, T = h->T;
    b.B = B; b.mode = mode;
    b.x4 = a.take<float>((size_t)B * P * 4);
    b.V = a.take<float>((size_t)N * 9 * Cd);
    b.dec_out = a.take<float>((size_t)N * P * 4);
    b.g = a.take<float>((size_t)N * P * 4);
    b.part = a.take<double>((size_t)B * pixel_blocks_per_image(P) * (6 * h->K + 3));
    b.lnstat = a.take<float>((size_t)N * 8);
    b.ll_img = a.take<float>((size_t)B);
    b.img_terms = a.take<float>((size_t)(T + 1) * B * 2);
    b.scal = a.take<float>((size_t)(T + 1) * 3 + 4);
    b.rows = a.take<float>((size_t)N * h->S * 3 * Cd);
    b.rows_p = a.take<float>((size_t)N * h->S * (h->S / 16 > 0 ? h->S / 16 : 1) * (mode == 1 ? 4 : 3) * Cd);   // per-tile row sums (EPI_L0ROWS / EPI_L0ROWSX)
    b.l0scr = a.take<float>(l0_rows_scratch_floats(N, Cd));
    b.Rc = a.take<float>((size_t)N * 9 * Cd);
    b.pm = a.take<float>((size_t)N * L);
    b.plv = a.take<float>((size_t)N * L);
    b.act.resize(h->Dd);
    for (int l = 0; l < h->Dd; ++l) b.act[l] = a.take<float>((size_t)N * P * Cd);
    b.dpre[0] = a.take<float>((size_t)N * P * Cd);
    b.dpre[1] = a.take<float>((size_t)N * P * Cd);
    b.tmax_act.resize(h->Dd);
    for (int l = 0; l < h->Dd; ++l) b.tmax_act[l] = a.take<float>(conv_ws_tmax_floats(N, h->S));
    b.tmax_dpre[0] = a.take<float>(conv_ws_tmax_floats(N, h->S));
    b.tmax_dpre[1] = a.take<float>(conv_ws_tmax_floats(N, h->S));
    const int ncopy = mode == 1 ? T + 1 : 1;
    auto per_iter = [&](std::vector<float*>& v, size_t n, int copies) {      // `copies` instances back to back, the rest alias [0]
        v.resize(T + 1);
        float* base = a.take<float>(n * (size_t)copies);
        for (int i = 0; i <= T; ++i) v[i] = (i < copies && base) ? base + (size_t)i * n : base;
    };
    per_iter(b.z, (size_t)N * L, ncopy);
    if (mode != 1) b.z[T] = a.take<float>((size_t)N * L);   // the final sample must not overwrite the last elbo()'s z (self.z)
    per_iter(b.g_pm, (size_t)N * L, ncopy);
    per_iter(b.g_plv, (size_t)N * L, ncopy);
    per_iter(b.latent, (size_t)N * 4 * L, ncopy);
    {
        // training keeps the refinement inputs of all T iterations, back to back: the conv stack of the refinement network is
        // back-propagated for all iterations in ONE batch of T * N slot-images (iodine_train_backward)
        const size_t n_enc = (size_t)N * P * 20;
        float* base = a.take<float>(n_enc * (mode == 1 ? T : 1));
        b.enc.resize(T + 1);
        for (int i = 0; i <= T; ++i) b.enc[i] = (mode == 1 && i < T) ? (base ? base + (size_t)i * n_enc : nullptr) : base;
        // split form (refine_split): the same memory as per-slot tensors [N][P][12] of all kept iterations back to back,
        // followed by the per-image tensors [B][P][8]
        const size_t n_k = (size_t)N * P * 12, n_s = (size_t)B * P * 8;
        const int keep = mode == 1 ? T : 1;
        b.enck.resize(T + 1); b.encs.resize(T + 1);
        for (int i = 0; i <= T; ++i) {
            const int j = (mode == 1 && i < T) ? i : 0;
            b.enck[i] = base ? base + (size_t)j * n_k : nullptr;
            b.encs[i] = base ? base + (size_t)keep * n_k + (size_t)j * n_s : nullptr;
        }
        b.rmap = a.take<float>((size_t)B * (h->S / 2) * (h->S / 2) * Cr);
    }
    per_iter(b.pooled, (size_t)N * Cr, ncopy);
    per_iter(b.u, (size_t)N * H, ncopy);
    per_iter(b.gates, (size_t)N * 4 * H, ncopy);
    per_iter(b.xin, (size_t)N * (H + 4 * L), ncopy);
    {   // gate GEMM of the refinement head (head_mfma): its input rows [u | latent | h_prev] and its output, rows padded to a multiple of 32
        const size_t Np = ((size_t)N + 31) / 32 * 32;
        b.head_xh = a.take<float>(Np * (size_t)(H + 4 * L + H));
        b.head_gp = a.take<float>(Np * (size_t)4 * H);
    }
    // LSTM state: h[i], c[i] = state BEFORE iteration i; inference ping-pongs two copies
    b.h.resize(T + 2); b.c.resize(T + 2);
    if (mode == 1) {
        // (each array contiguous over the iterations: the head's weight gradients are one GEMM over all T * N rows)
        float* hb = a.take<float>((size_t)(T + 2) * N * H);
        float* cb = a.take<float>((size_t)(T + 2) * N * H);
        for (int i = 0; i <= T + 1; ++i) { b.h[i] = hb ? hb + (size_t)i * N * H : nullptr; b.c[i] = cb ? cb + (size_t)i * N * H : nullptr; }
    } else {
        float* hh[2] = {a.take<float>((size_t)N * H), a.take<float>((size_t)N * H)};
        float* cc[2] = {a.take<float>((size_t)N * H), a.take<float>((size_t)N * H)};
        for (int i = 0; i <= T + 1; ++i) { b.h[i] = hh[i & 1]; b.c[i] = cc[i & 1]; }
    }
    b.ract.assign(T + 1, std::vector<float*>(h->Dr));
    {
        int s = h->S;
        for (int l = 0; l < h->Dr; ++l) {
            s = ref_out_size(h, s);
            const size_t n_l = (size_t)N * s * s * Cr;
            float* base = a.take<float>(n_l * (mode == 1 ? T : 1));         // [T][N][s][s][Cr] in training (see enc)
            for (int i = 0; i <= T; ++i) b.ract[i][l] = (mode == 1 && i < T) ? (base ? base + (size_t)i * n_l : nullptr) : base;
        }
    }
    if (mode == 1) {
        const int Cmax = Cd > Cr ? Cd : Cr;
        const size_t part_elems = (size_t)512 * 4 * 9 * 32 * 32 > (size_t)512 * 9 * Cmax * Cmax
                                      ? (size_t)512 * 4 * 9 * 32 * 32 : (size_t)512 * 9 * Cmax * Cmax;
        b.wg_part = a.take<float>(part_elems);
        b.wg_part_b = a.take<float>((size_t)512 * 64);
        b.wg_acc.assign(h->Dd, nullptr); b.wg_acc_b.assign(h->Dd, nullptr);
        if (h->wgrad_accum && !h->generic) {
            b.wg_acc[0] = a.take<float>((size_t)1024 * 2 * 9 * Cd * 4);       // dec_out_bwd_fused: <= 1024 blocks x KS <= 2 tiles of [9][Cd][4]
            b.wg_acc_b[0] = a.take<float>((size_t)1024 * 4);
            for (int l = 1; l < h->Dd; ++l) {
                b.wg_acc[l] = a.take<float>((size_t)512 * 4 * 9 * 32 * 32 > (size_t)512 * 9 * Cd * Cd ? (size_t)512 * 4 * 9 * 32 * 32 : (size_t)512 * 9 * Cd * Cd);
                b.wg_acc_b[l] = a.take<float>((size_t)512 * 64);
            }
        }
        b.wg_fold = a.take<float>((size_t)WGRAD_FOLD * 9 * Cmax * Cmax);
        b.Dsum = a.take<float>((size_t)P * Cd);
        b.Dpart = a.take<float>((size_t)l0_dgroups(N) * P * Cd);
        b.rown = a.take<float>((size_t)N * h->S * 4 * Cd);
        b.Rsum = a.take<float>((size_t)h->S * 4 * Cd);
        b.RT = a.take<float>((size_t)N * 9 * Cd);
        b.tmp_lz = a.take<float>((size_t)L * 9 * Cd);
        // ddm / ddv / dgates / ds: one instance per iteration (weight gradients of the head in one pass over all of them)
        b.ddm = a.take<float>((size_t)T * N * L); b.ddv = a.take<float>((size_t)T * N * L);
        b.dc1 = a.take<float>((size_t)N * H); b.dgates = a.take<float>((size_t)T * N * 4 * H);
        b.dxin = a.take<float>((size_t)N * H); b.ds = a.take<float>((size_t)T * N * H);
        b.dpooled = a.take<float>((size_t)T * N * Cr);                     // all iterations (batched conv-stack backward)
        for (int j = 0; j < 2; ++j) { b.carry_h[j] = a.take<float>((size_t)N * H); b.carry_c[j] = a.take<float>((size_t)N * H); }
        b.rdpre.resize(h->Dr);
        int s = h->S;
        for (int l = 0; l < h->Dr; ++l) { s = ref_out_size(h, s); b.rdpre[l] = a.take<float>((size_t)T * N * s * s * Cr); }
    }
    if (h->generic) {
        b.gen_l0 = a.take<float>(gen_l0_scratch_floats(N, h->S, Cd, h->kd));   // row / tap sums, prefix table of the broadcast layer
    }
    if ((h->generic || h->gen_ref) && mode == 1) {
        size_t scr = 0;
        if (h->generic) scr = std::max(gen_wgrad_scratch_floats(Cd, 4, h->kd), gen_wgrad_scratch_floats(Cd, Cd, h->kd));
        if (h->gen_ref) scr = std::max(scr, std::max(gen_wgrad_scratch_floats(17, Cr, h->kr), gen_wgrad_scratch_floats(Cr, Cr, h->kr)));
        b.gen_scr = a.take<float>(scr);
    }
    b.bytes = (a.off + 255) & ~(size_t)255;
}

void drop_graphs(iodine_handle* h)
{
    for (auto& g : h->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    h->graphs.clear();
    h->seen_keys.clear();
}

int ensure_workspace(iodine_handle* h, int B, int mode)
{
    if (h->buf.B == B && h->buf.mode == mode && h->buf.bytes > 0) return IODINE_OK;
    Arena q(nullptr); Buffers tmp; plan(h, B, mode, q, tmp);
    void* base = nullptr;
    if (h->ws_user) {
        if (h->ws_user_bytes < tmp.bytes) {
            char m[160];
            snprintf(m, sizeof m, "workspace too small: need %zu bytes for batch %d mode %d, have %zu", tmp.bytes, B,
                     mode, h->ws_user_bytes);
            return h->fail(IODINE_ERR_WORKSPACE, m);
        }
        base = h->ws_user;
    } else {
        if (h->ws_own_bytes < tmp.bytes) {
            if (h->ws_own) { HIPCHK(h, hipFree(h->ws_own)); h->ws_own = nullptr; h->ws_own_bytes = 0; }
            HIPCHK(h, hipMalloc(&h->ws_own, tmp.bytes));
            h->ws_own_bytes = tmp.bytes;
        }
        base = h->ws_own;
    }
    Arena a(base);
    plan(h, B, mode, a, h->buf);
    h->fwd_done = false;                 // a re-planned arena no longer holds the saved forward / the last elbo() outputs
    h->last_elbo_iter = -1;
    h->enc_valid = false;
    // captured graphs stay: their key holds the arena's base address, the batch and (through the entry point) the mode, and the
    // carve-up is a pure function of those - a step that alternates training and reconstruct calls keeps replaying both
    return IODINE_OK;
}

// one decoder forward pass from z (already in buf.V via dec_v) -> dec_out
int decoder_forward(iodine_handle* h, hipStream_t st, int N, const float* z, float* out = nullptr)
{
    Buffers& b = h->buf;
    if (!out) out = b.dec_out;
    if (h->generic) {
        // fallback: the broadcast layer from the prefix table of its per-tap latent products (kernels_genl0.hip: the broadcast tensor is
        // never built), every other layer a generic fp32 conv (kernels_generic.hip)
        PROF(h, st, "gen_l0", launch_gen_l0_fwd(st, z, h->gen_wdec[0], h->gen_cterm, b.gen_l0, b.act[0], N, h->L, h->S, h->Cd, h->kd));
        const float* in = b.act[0];
        for (int l = 1; l < h->Dd; ++l) {
            PROF(h, st, "gen_conv", launch_gen_conv_fwd(st, in, h->gen_wdec[l], h->dec_b[l], b.act[l], N, h->S, h->Cd, h->Cd, h->Cd, h->kd, 1, 1));
            in = b.act[l];
        }
        PROF(h, st, "gen_conv", launch_gen_conv_fwd(st, in, h->gen_wout, h->dec_out_b, out, N, h->S, h->Cd, h->Cd, 4, h->kd, 1, 0));
        return IODINE_OK;
    }
    const bool ws = conv_ws_ok(h);
    PROF(h, st, "dec_l0", launch_dec_l0(st, b.V, h->cmap, b.act[0], N, h->S, h->Cd, ws ? b.tmax_act[0] : nullptr));
    for (int l = 1; l < h->Dd; ++l) {
        if (h->precision == 1)
            PROF(h, st, "conv_tile_fwd", conv_f16x3(h, st, b.act[l - 1], h->dec_wf16[l], h->dec_wsf[l], h->dec_wmeta[l], h->dec_b[l],
                                                    nullptr, b.act[l], b.tmax_act[l - 1], b.tmax_act[l], N, h->S, h->Cd, h->Cd,
                                                    EPI_BIAS_ELU, l));
        else if (conv_ws32_ok(h))
            PROF(h, st, "conv_tile_fwd", launch_conv3x3_ws_f32(st, b.act[l - 1], h->dec_wsf[l], h->dec_b[l], nullptr, b.act[l], N, h->S, h->Cd,
                                                               EPI_BIAS_ELU, l & 1));
        else
            PROF(h, st, "conv_tile_fwd", launch_conv3x3_tile(st, b.act[l - 1], h->dec_wf[l], h->dec_b[l], nullptr,
                                                             b.act[l], N, h->S, h->Cd, h->Cd, EPI_BIAS_ELU));
    }
    if (h->precision == 1 && h->dec_out_rows && ws && dec_out_rows_ok(h->S, h->Cd, b.tmax_act[h->Dd - 1]))
        PROF(h, st, "dec_out", launch_dec_out_rows_f16x3(st, b.act[h->Dd - 1], h->dec_out_w16, h->dec_out_meta, h->dec_out_b, out, N, h->S, h->Cd,
                                                         b.tmax_act[h->Dd - 1]));
    else if (h->precision == 1)
        PROF(h, st, "dec_out", launch_dec_out_stream_f16x3(st, b.act[h->Dd - 1], h->dec_out_w16, h->dec_out_meta, h->dec_out_b, out, N, h->S,
                                                           h->Cd, ws ? b.tmax_act[h->Dd - 1] : nullptr));
    else if (h->dec_out_rows && dec_out_rows_ok(h->S, h->Cd, b.act[h->Dd - 1]) && h->dec_out_w32)      // exact fp32 MFMA, row-streaming
        PROF(h, st, "dec_out", launch_dec_out_rows_f16x3(st, b.act[h->Dd - 1], h->dec_out_w32, nullptr, h->dec_out_b, out, N, h->S, h->Cd, nullptr, 1));
    else
        PROF(h, st, "dec_out", launch_dec_out(st, b.act[h->Dd - 1], h->dec_out_w, h->dec_out_b, out, N, h->S, h->Cd));
    return IODINE_OK;
}

// reduce the partial tiles of the last wgrad launch into the accumulators of (weight, bias)
int reduce_wgrad(iodine_handle* h, hipStream_t st, int nparts, int ci_pad, int co_pad, int O_real, int I_real,
                 int I_dst, float alpha, int wparam, int bparam, int nbias_parts)
{
    Buffers& b = h->buf;
    HIPCHK(h, launch_wgrad_reduce(st, b.wg_part, nparts, ci_pad, co_pad, O_real, I_real, I_dst, alpha, h->gacc[wparam], b.wg_fold,
                                  b.wg_part_b, nbias_parts, h->gacc[bparam]));
    return IODINE_OK;
}

// gradient of B*ELBO wrt the decoder input z through the whole decoder (replaces the autograd traversal of
// (B*elbo).backward(), iodine.py:90,137).  Leaves d(pre-activation) of layer 0 in the returned buffer.
// With train_alpha != 0 the decoder weight gradients of this pass are accumulated on the way with that factor
// (= -w_i / B): they are what the outer loss.backward() (train.py:63) would compute for this decoder pass.
// the same on the generic fallback path: plain chain of data gradients (and, in training, weight gradients with the pass factor);
// the broadcast layer's weight gradient and the gradient wrt z come from the tap-window sums of its pre-activation gradient
// (kernels_genl0.hip); dz is left in the first L entries of every row of buf.Rc (dz_latent multiplies that by the identity in h->gen_ident)
int decoder_backward_generic(iodine_handle* h, hipStream_t st, int N, float train_alpha, int it)
{
    Buffers& b = h->buf;
    const int Cd = h->Cd, Dd = h->Dd, L = h->L, S = h->S, k = h->kd;
    auto G = [&](const std::string& name) { return h->gacc[param_index(h, name)]; };
    int cur = 0;
    PROF(h, st, "gen_conv", launch_gen_conv_dgrad(st, b.g, h->gen_wout, b.act[Dd - 1], b.dpre[cur], N, S, Cd, Cd, 4, k, 1));
    if (train_alpha != 0.f)
        PROF(h, st, "gen_conv", launch_gen_conv_wgrad(st, b.act[Dd - 1], b.g, b.gen_scr, N, S, Cd, Cd, Cd, 4, k, 1, train_alpha,
                                                      G("decoder.conv.weight"), G("decoder.conv.bias")));
    for (int l = Dd - 1; l > 0; --l) {
        const std::string base = "decoder.mlc.layers." + std::to_string(l);
        if (train_alpha != 0.f)
            PROF(h, st, "gen_conv", launch_gen_conv_wgrad(st, b.act[l - 1], b.dpre[cur], b.gen_scr, N, S, Cd, Cd, Cd, Cd, k, 1, train_alpha,
                                                          G(base + ".weight"), G(base + ".bias")));
        PROF(h, st, "gen_conv", launch_gen_conv_dgrad(st, b.dpre[cur], h->gen_wdec[l], b.act[l - 1], b.dpre[cur ^ 1], N, S, Cd, Cd, Cd, k, 1));
        cur ^= 1;
    }
    HIPCHK(h, hipMemsetAsync(b.Rc, 0, sizeof(float) * (size_t)N * 9 * Cd, st));
    PROF(h, st, "gen_l0", launch_gen_l0_bwd(st, b.dpre[cur], b.z[it], h->gen_wdec[0], h->lin, b.gen_l0, N, L, S, Cd, k, train_alpha,
                                            G("decoder.mlc.layers.0.weight"), G("decoder.mlc.layers.0.bias"), b.Rc, 9 * Cd));
    return IODINE_OK;
}

int decoder_backward_data(iodine_handle* h, hipStream_t st, int N, float** dpre0, float train_alpha, int it)
{
    Buffers& b = h->buf;
    const int Cd = h->Cd, Dd = h->Dd;
    if (h->generic) { *dpre0 = nullptr; return decoder_backward_generic(h, st, N, train_alpha, it); }
    int cur = 0, nparts = 0, ncop = 0, nb = 0, rc;
    bool fused_l0 = false;
    // training: one pass over the last hidden activation gives the data gradient AND the weight / bias gradient
    const bool out_fused = train_alpha != 0.f && h->precision == 1 && h->out_bwd_fused;
    // round 5: a block's partial tile accumulates alpha_i x (pass i) in a per-layer buffer; reduced once, after the last pass (it == T)
    const bool acc_w = train_alpha != 0.f && h->wgrad_accum && !b.wg_acc.empty() && b.wg_acc[0];
#ifdef IODINE_XSKIP_HOOK
    if (!(g_iod_xskip & 256))
#endif
    if (out_fused) {
        const int wi = param_index(h, "decoder.conv.weight"), bi = param_index(h, "decoder.conv.bias");
        if (acc_w) {
            PROF(h, st, "dec_out_bwd", launch_dec_out_bwd_fused_f16x3(st, b.act[Dd - 1], b.g, h->dec_out_wb16, h->dec_out_meta,
                                                                       b.dpre[cur], conv_ws_ok(h) ? b.tmax_dpre[cur] : nullptr,
                                                                       b.wg_acc[0], b.wg_acc_b[0], N, h->S, Cd, &nparts, &nb, train_alpha, it != 0));
            if (it == h->T)
                HIPCHK(h, launch_wgrad_reduce(st, b.wg_acc[0], nparts, Cd, 4, 4, Cd, Cd, 1.f, h->gacc[wi], b.wg_fold, b.wg_acc_b[0], nb, h->gacc[bi]));
        } else {
        PROF(h, st, "dec_out_bwd", launch_dec_out_bwd_fused_f16x3(st, b.act[Dd - 1], b.g, h->dec_out_wb16, h->dec_out_meta,
                                                                   b.dpre[cur], conv_ws_ok(h) ? b.tmax_dpre[cur] : nullptr,
                                                                   b.wg_part, b.wg_part_b, N, h->S, Cd, &nparts, &nb));
        HIPCHK(h, launch_wgrad_reduce(st, b.wg_part, nparts, Cd, 4, 4, Cd, Cd, train_alpha, h->gacc[wi], b.wg_fold,
                                      b.wg_part_b, nb, h->gacc[bi]));
        }
    } else
    {
        if (h->precision == 1)
            PROF(h, st, "dec_out_dgrad", launch_dec_out_dgrad_f16x3(st, b.g, h->dec_out_wb16, h->dec_out_meta, b.act[Dd - 1],
                                                                     b.dpre[cur], N, h->S, Cd, conv_ws_ok(h) ? b.tmax_dpre[cur] : nullptr));
        else {
            PROF(h, st, "dec_out_dgrad", launch_conv3x3_tile(st, b.g, h->dec_out_wb, nullptr, b.act[Dd - 1], b.dpre[cur], N,
                                                             h->S, 4, Cd, EPI_MUL_ELUGRAD));
            // the generic kernel leaves no per-cell max for the weight-stationary conv that consumes its output
            if (conv_ws_ok(h)) HIPCHK(h, launch_cell_max(st, b.dpre[cur], b.tmax_dpre[cur], N, h->S, Cd));
        }
    }
    if (train_alpha != 0.f && !out_fused) {
        const int wi = param_index(h, "decoder.conv.weight"), bi = param_index(h, "decoder.conv.bias");
        if (h->precision == 1) {                           // GEMM form: rows (tap, co), no N = 4 -> 32 padding
            PROF(h, st, "dec_out_wgrad", launch_dec_out_wgrad_gemm_f16x3(st, b.act[Dd - 1], b.g, b.wg_part, b.wg_part_b, N, h->S,
                                                                          Cd, &nparts, &nb));
            HIPCHK(h, launch_wgrad_reduce(st, b.wg_part, nparts, Cd, 4, 4, Cd, Cd, train_alpha, h->gacc[wi], b.wg_fold,
                                          b.wg_part_b, nb, h->gacc[bi]));
        } else if (conv_ws32_ok(h)) {                      // exact fp32, GEMM form (kernels_wgrad32.hip)
            PROF(h, st, "dec_out_wgrad", launch_dec_out_wgrad_f32(st, b.act[Dd - 1], b.g, b.wg_part, b.wg_part_b, N, h->S, Cd, &nparts, &nb));
            HIPCHK(h, launch_wgrad_reduce(st, b.wg_part, nparts, Cd, 4, 4, Cd, Cd, train_alpha, h->gacc[wi], b.wg_fold,
                                          b.wg_part_b, nb, h->gacc[bi]));
        } else {
            PROF(h, st, "dec_out_wgrad", launch_conv3x3_wgrad_tile(st, b.act[Dd - 1], b.g, b.wg_part, b.wg_part_b, N,
                                                                    h->S, Cd, 4, &nparts, &ncop, &nb));
            rc = reduce_wgrad(h, st, nparts, Cd, ncop, 4, Cd, Cd, train_alpha, wi, bi, nb);
            if (rc) return rc;
        }
    }
    for (int l = Dd - 1; l >= 1; --l) {
        if (train_alpha != 0.f && acc_w && (h->precision == 1 || conv_ws32_ok(h))) {
            if (h->precision == 1)
                PROF(h, st, "conv_tile_wgrad", launch_conv3x3_wgrad_f16x3_ws(st, b.act[l - 1], b.dpre[cur], b.wg_acc[l], b.wg_acc_b[l], N, h->S, Cd, Cd,
                                                                              &nparts, &ncop, &nb, train_alpha, it != 0));
            else
                PROF(h, st, "conv_tile_wgrad", launch_conv3x3_wgrad_f32_ws(st, b.act[l - 1], b.dpre[cur], b.wg_acc[l], b.wg_acc_b[l], N, h->S, Cd,
                                                                            &nparts, &ncop, &nb, train_alpha, it != 0));
            if (it == h->T) {
                const std::string base = "decoder.mlc.layers." + std::to_string(l);
                HIPCHK(h, launch_wgrad_reduce(st, b.wg_acc[l], nparts, Cd, ncop, Cd, Cd, Cd, 1.f, h->gacc[param_index(h, base + ".weight")], b.wg_fold,
                                              b.wg_acc_b[l], nb, h->gacc[param_index(h, base + ".bias")]));
            }
        } else if (train_alpha != 0.f) {
            if (h->precision == 1)
                PROF(h, st, "conv_tile_wgrad", launch_conv3x3_wgrad_f16x3_ws(st, b.act[l - 1], b.dpre[cur], b.wg_part,
                                                                              b.wg_part_b, N, h->S, Cd, Cd, &nparts, &ncop, &nb));
            else if (conv_ws32_ok(h))
                PROF(h, st, "conv_tile_wgrad", launch_conv3x3_wgrad_f32_ws(st, b.act[l - 1], b.dpre[cur], b.wg_part, b.wg_part_b, N, h->S, Cd,
                                                                            &nparts, &ncop, &nb));
            else
                PROF(h, st, "conv_tile_wgrad", launch_conv3x3_wgrad_tile(st, b.act[l - 1], b.dpre[cur], b.wg_part,
                                                                          b.wg_part_b, N, h->S, Cd, Cd, &nparts, &ncop, &nb));
            const std::string base = "decoder.mlc.layers." + std::to_string(l);
            rc = reduce_wgrad(h, st, nparts, Cd, ncop, Cd, Cd, Cd, train_alpha, param_index(h, base + ".weight"),
                              param_index(h, base + ".bias"), nb);
            if (rc) return rc;
        }
        // Inference: nothing but the broadcast layer's row / class sums needs d(pre-activation 0), so the last data gradient
        // reduces its tile to per-row sums in its epilogue (EPI_L0ROWS) and the 0.94 GB tensor is neither written nor re-read.
        // Training: the same with one more sum per row (EPI_L0ROWSX, weight-stationary kernel only): class sums for dz and the
        // latent-channel weights, slot-summed row sums for the coordinate-channel weights and the bias.
        fused_l0 = l == 1 && h->fuse_l0 && (h->precision == 1 ? (train_alpha == 0.f || conv_ws_ok(h)) : conv_ws32_ok(h));
        if (h->precision == 1)
            PROF(h, st, "conv_tile_dgrad", conv_f16x3(h, st, b.dpre[cur], h->dec_wb16[l], h->dec_wsb[l], h->dec_wmeta[l] + 2, nullptr,
                                                      b.act[l - 1], fused_l0 ? b.rows_p : b.dpre[cur ^ 1], b.tmax_dpre[cur],
                                                      b.tmax_dpre[cur ^ 1], N, h->S, Cd, Cd,
                                                      fused_l0 ? (train_alpha != 0.f ? EPI_L0ROWSX : EPI_L0ROWS) : EPI_MUL_ELUGRAD, l));
        else if (conv_ws32_ok(h))
            PROF(h, st, "conv_tile_dgrad", launch_conv3x3_ws_f32(st, b.dpre[cur], h->dec_wsb[l], nullptr, b.act[l - 1],
                                                                 fused_l0 ? b.rows_p : b.dpre[cur ^ 1], N, h->S, Cd,
                                                                 fused_l0 ? (train_alpha != 0.f ? EPI_L0ROWSX : EPI_L0ROWS) : EPI_MUL_ELUGRAD, l & 1));
        else
            PROF(h, st, "conv_tile_dgrad", launch_conv3x3_tile(st, b.dpre[cur], h->dec_wb[l], nullptr, b.act[l - 1],
                                                               b.dpre[cur ^ 1], N, h->S, Cd, Cd, EPI_MUL_ELUGRAD));
        cur ^= 1;
    }
    *dpre0 = b.dpre[cur];
    if (fused_l0 && train_alpha == 0.f) {
        PROF(h, st, "l0_reduce", launch_l0_reduce_cls_tiles(st, b.rows_p, b.Rc, N, h->S, Cd, b.l0scr));
        return IODINE_OK;
    }
    if (fused_l0) {
        PROF(h, st, "l0_reduce", launch_l0_reduce_cls_tiles_x(st, b.rows_p, b.Rc, b.rown, N, h->S, Cd, b.l0scr));
        HIPCHK(h, launch_l0_rowsum_acc(st, b.rown, N, h->S, Cd, train_alpha, it == 0, b.Rsum));
    } else
    // row / class sums of dpre0 for dz; in training the same read also feeds the slot-summed gradient map, which is
    // accumulated (with this pass's factor) over the T+1 passes and consumed once after the last one
    PROF(h, st, "l0_reduce", launch_l0_reduce(st, *dpre0, b.rows, b.Rc, N, h->S, Cd, train_alpha != 0.f ? b.Dpart : nullptr,
                                              b.Dsum, train_alpha, it == 0));
    if (train_alpha != 0.f) {
        // layer 0 (spatial broadcast): latent-channel weights from z and the per-tap sums, coordinate channels
        // and bias from the slot-summed gradient map
        const int wi = param_index(h, "decoder.mlc.layers.0.weight"), bi = param_index(h, "decoder.mlc.layers.0.bias");
        if (sgemm_tn_mfma_ok(h->L, 9 * Cd, N)) {            // z^T . RT on fp32 MFMA, accumulated straight into gw[co][ci][tap]
            HIPCHK(h, launch_l0_tap_sums(st, b.Rc, b.RT, N, Cd));
            HIPCHK(h, launch_sgemm_tn_mfma(st, h->L, 9 * Cd, N, train_alpha, b.z[it], h->L, b.RT, 9 * Cd, 1.f, h->gacc[wi], h->L + 2, 1, Cd));
        } else                                              // (round 6) tap sums + product + scatter in one launch
            HIPCHK(h, launch_l0_latent_wgrad(st, b.Rc, b.z[it], N, h->L, Cd, train_alpha, h->gacc[wi]));
        if (it == h->T) {
            if (fused_l0) HIPCHK(h, launch_l0_coord_grads_rows(st, b.Rsum, h->lin, h->S, Cd, h->L, 1.f, h->gacc[wi], h->gacc[bi]));
            else HIPCHK(h, launch_l0_coord_grads(st, b.Dsum, h->lin, h->S, Cd, h->L, 1.f, h->gacc[wi], h->gacc[bi], b.wg_part));
        }
    }
    return IODINE_OK;
}

// elbo() + inner backward + get_input_encoding for iteration i (iodine.py:85-93 / 133-142)
int elbo_and_gradients(iodine_handle* h, hipStream_t st, int B, const float* eps_i, int i, bool need_grads,
                       float train_alpha = 0.f)
{
    Buffers& b = h->buf;
    const int N = B * h->K;
    HIPCHK(h, launch_dec_v(st, b.pm, b.plv, eps_i, nullptr, h->wcls, b.z[i], b.V, N, h->L, h->Cd));
    int rc = decoder_forward(h, st, N, b.z[i]);
    if (rc) return rc;
    PROF(h, st, "pixel_pass1", launch_pixel_pass1(st, b.x4, b.dec_out, b.g, b.part, B, h->K, h->P, (float)h->cfg.sigma, h->precision == 0));
    // the ticket of pixel_finalize_elbo_kernel is reset by the last block of every launch; the first launch of an entry point also
    // starts from a fresh 0 (a memset node under graph capture), whatever a failed call or a misuse of the handle from a second
    // stream left in it - once per call, not per launch (a memset is a launch of its own)
    if (i == 0) HIPCHK(h, hipMemsetAsync(h->elbo_counter, 0, sizeof(unsigned), st));
    HIPCHK(h, launch_pixel_finalize_elbo(st, b.part, B, h->K, h->P, h->cfg.layernorm, b.lnstat, b.ll_img, b.pm, b.plv, h->L,
                                         b.img_terms + (size_t)i * B * 2, b.scal + 3 * i, h->elbo_counter));
    h->last_elbo_iter = i; h->last_elbo_batch = B;
    if (!need_grads) return IODINE_OK;
    float* dpre0 = nullptr;
    rc = decoder_backward_data(h, st, N, &dpre0, train_alpha, i);
    if (rc) return rc;
    HIPCHK(h, launch_dz_latent(st, b.Rc, h->generic ? h->gen_ident : h->wclsT, b.pm, b.plv, eps_i, N, h->L, h->Cd, h->cfg.layernorm,
                               b.g_pm[i], b.g_plv[i], b.latent[i], h->Lreal));
    return IODINE_OK;
}

bool refine_split_on(const iodine_handle* h);

bool refine_f16_ok(const iodine_handle* h)
{
    if (h->gen_ref || (h->Cr != 64 && h->Cr != 32)) return false;
    int s = h->S;
    for (int l = 0; l < h->Dr; ++l) { if (s % 2 != 0) return false; s /= 2; }
    return true;
}

// (round 5: the exact-fp32 path runs the same tuned stride-2 kernels in their fp32-MFMA form, so it takes the split first layer too)
bool refine_split_on(const iodine_handle* h) { return h->refine_split && refine_f16_ok(h); }

// refine() + posterior.update() for iteration i (iodine.py:95-100 / 144-145)
int refine_step(iodine_handle* h, hipStream_t st, int B, int i, bool save)
{
    Buffers& b = h->buf;
    const int N = B * h->K;
    // split first layer: the channels every slot of an image shares are written and convolved once per image
    const bool split = refine_split_on(h);     // (training: iodine_train_forward records the form in h->fwd_split - host state must
                                               //  not be written here, a hipGraph replay does not execute this body)
    // ... and with refine_l0_fused the encoding is not written at all (inference): one kernel from the decoder output to layer 0's output
    const bool l0f = split && h->precision == 1 && h->refine_l0_fused && refine_l0_fused_ok(h->S, h->Cr, h->K);
    const int f32 = h->precision == 0;         // exact fp32 products: the fp32-MFMA form of the same stride-2 kernels
    const bool keep_enc = save || h->stop_after >= 0;     // the backward / iodine_debug_copy("enc") read it
    if (l0f)
        PROF(h, st, "refine_l0f", launch_refine_l0_fused(st, b.x4, b.dec_out, b.lnstat, h->lin, h->ref_l0k, h->ref_l0kmeta, h->ref_l0s,
                                                         h->ref_l0smeta, h->ref_b[0], b.ract[i][0], keep_enc ? b.enck[i] : nullptr,
                                                         keep_enc ? b.encs[i] : nullptr, B, h->K, h->S, h->Cr, (float)h->cfg.sigma,
                                                         h->enc_chmask));
    else
        PROF(h, st, "pixel_pass2", launch_pixel_pass2(st, b.x4, b.dec_out, b.lnstat, h->lin, split ? b.enck[i] : b.enc[i], B, h->K, h->S,
                                                      (float)h->cfg.sigma, split ? b.encs[i] : nullptr, h->enc_chmask, h->precision == 0));
    int s = h->S;
    const float* in = b.enc[i];
    for (int l = 0; l < h->Dr; ++l) {
        if (h->gen_ref) {
            PROF(h, st, "gen_conv", launch_gen_conv_fwd(st, in, h->gen_wref[l], h->ref_b[l], b.ract[i][l], N, s, l == 0 ? 17 : h->Cr,
                                                        l == 0 ? 20 : h->Cr, h->Cr, h->kr, h->rs, 1, l == 0 ? h->enc_chmask : 0xffffffffu));
        } else if (l == 0 && l0f) {
        } else if (l == 0 && split) {
            PROF(h, st, "refine_l0", launch_conv3x3_s2_f16x3(st, b.encs[i], h->ref_wsh16, h->ref_wshmeta, nullptr, b.rmap, B, s, 8,
                                                             h->Cr, nullptr, 0, f32));
            PROF(h, st, "refine_l0", launch_conv3x3_s2_f16x3(st, b.enck[i], h->ref_wk16, h->ref_wkmeta, h->ref_b[0], b.ract[i][0], N,
                                                             s, 12, h->Cr, b.rmap, h->K, f32));
        } else if (l > 0 && h->refine_ws && refine_f16_ok(h) && conv3x3_s2ws_ok(s, h->Cr)) {
            PROF(h, st, "refine_conv", launch_conv3x3_s2ws_f16x3(st, in, h->ref_wsf[l], h->ref_wsf_meta[l], h->ref_b[l], b.ract[i][l], N, s, h->Cr, f32));
        } else if (refine_f16_ok(h))
            PROF(h, st, l == 0 ? "refine_l0" : "refine_conv", launch_conv3x3_s2_f16x3(st, in, h->ref_wf16[l], h->ref_wmeta[l], h->ref_b[l],
                                                               b.ract[i][l], N, s, l == 0 ? 20 : h->Cr, h->Cr, nullptr, 0, f32));
        else
            PROF(h, st, l == 0 ? "refine_l0" : "refine_conv", launch_conv3x3_gather(st, in, h->ref_w[l], h->ref_b[l], b.ract[i][l], N, s, s,
                                                             l == 0 ? 20 : h->Cr, h->Cr, 2));
        in = b.ract[i][l];
        s = ref_out_size(h, s);
    }
    PROF(h, st, "refine_head", launch_refine_head(st, in, N, s * s, h->Cr, h->H, h->L, h->mlp_wT, h->mlp_b, h->wihT, h->whhT, h->lstm_b,
                                 h->wmT, h->bm, h->wvT, h->bv, b.latent[i], b.h[i], b.c[i], b.h[i + 1], b.c[i + 1], b.pm,
                                 b.plv, save ? b.pooled[i] : nullptr, save ? b.u[i] : nullptr,
                                 save ? b.gates[i] : nullptr, save ? b.xin[i] : nullptr, nullptr, nullptr,
                                 h->head_mfma ? b.head_xh : nullptr, h->head_mfma ? b.head_gp : nullptr));
    return IODINE_OK;
}

int check_ready(iodine_handle* h, int batch)
{
    if (!h) return IODINE_ERR_INVALID;
    if (!h->params_set) return h->fail(IODINE_ERR_STATE, "iodine_set_params has not been called");
    if (batch < 1) return h->fail(IODINE_ERR_INVALID, "batch must be >= 1");
    // several kernels index one activation tensor [N][P][C] with 32-bit element offsets
    const size_t cmax = (size_t)std::max(std::max(h->Cd, h->Cr), 20);
    if ((size_t)batch * h->K * h->P * cmax >= ((size_t)1 << 31))
        return h->fail(IODINE_ERR_INVALID, "batch too large for one device: batch * slots * pixels * channels must stay below 2^31 "
                                           "(shard the images over ranks, iodine_amd.parallel)");
    return IODINE_OK;
}

// Run `body` (a fixed-shape sequence of launches on `st`) eagerly, or - with option "graph" - through a hipGraph keyed by
// the full argument tuple: the first call with a tuple runs eagerly (it also performs the one-time hipFuncSetAttribute
// calls of the launchers), the second is captured + instantiated, later ones are a single hipGraphLaunch.  Host-side state
// changes must NOT live in `body` (a replay does not execute it).
template <typename F>
int run_graphed(iodine_handle* h, hipStream_t st, const std::vector<uintptr_t>& key, F&& body)
{
    if (!h->graph || h->profile) return body();
    if (!st) return h->fail(IODINE_ERR_INVALID, "option graph=1 needs a non-default stream (the legacy null stream cannot be captured)");
    for (auto& g : h->graphs)
        if (g.key == key) {
            g.used = ++h->graph_clock;
            HIPCHK(h, hipGraphLaunch(g.exec, st));
            ++h->graph_replays;
            return IODINE_OK;
        }
    bool seen = false;
    for (auto& k : h->seen_keys) if (k == key) { seen = true; break; }
    if (!seen) {
        if (h->seen_keys.size() >= 64) h->seen_keys.clear();
        h->seen_keys.push_back(key);
        return body();
    }
    HIPCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = body();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return h->fail(IODINE_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    GraphEntry ge;
    ge.key = key; ge.graph = graph;
    const hipError_t e2 = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    if (e2 != hipSuccess) { (void)hipGraphDestroy(graph); return h->fail(IODINE_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2)); }
    if (h->graphs.size() >= 8) {                           // least recently used out
        size_t lru = 0;
        for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].used < h->graphs[lru].used) lru = i;
        (void)hipGraphExecDestroy(h->graphs[lru].exec); (void)hipGraphDestroy(h->graphs[lru].graph);
        h->graphs.erase(h->graphs.begin() + lru);
    }
    ge.used = ++h->graph_clock;
    h->graphs.push_back(ge);
    ++h->graph_captures;
    HIPCHK(h, hipGraphLaunch(ge.exec, st));
    return IODINE_OK;
}

std::vector<uintptr_t> graph_key(const iodine_handle* h, int entry, int batch, std::initializer_list<const void*> ptrs)
{
    std::vector<uintptr_t> k = {(uintptr_t)entry, (uintptr_t)batch, (uintptr_t)h->stop_after, (uintptr_t)h->precision,
                                (uintptr_t)h->variant, (uintptr_t)h->fuse_l0, (uintptr_t)h->out_bwd_fused, (uintptr_t)h->refine_split, (uintptr_t)(h->head_fused | (h->refine_bwd_fused << 1) | (h->refine_ws << 2) | (h->refine_l0_fused << 3) | (h->head_mfma << 4) | (h->wgrad_accum << 5) | (h->dec_out_rows << 6)),
                                (uintptr_t)(h->ws_user ? h->ws_user : h->ws_own)};
    for (const void* p : ptrs) k.push_back((uintptr_t)p);
    return k;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6 - DIM_LATENT / REF.MLP_UNITS that are not multiples of 4 (the reference takes any: iodine.py:8-32, 446-464).
// The refinement-head kernels move weight rows as 16-byte vectors, so such a model runs on an INNER handle created at Lp = ceil4(L),
// Hp = ceil4(H) with every parameter zero-padded into the wider shapes.  Padding is exact, not approximate:
//   * padded latent entries have init_mean = init_logvar = 0, their decoder-input weights and the rows of mean_update / logvar_update
//     that produce them are 0, eps is padded with 0: z = mu = 0, logvar = 0 for the whole loop, KL contribution 1/2 (0 + 1 - 0 - 1) = 0,
//     d ELBO / d lambda = 0, and the LSTM reads them through zero columns of weight_ih;
//   * padded hidden units: zero MLP row and bias -> u = ELU(ELU(0)) = 0; zero gate rows and biases -> i = f = o = 1/2, g = 0 -> c1 = h1 = 0
//     from c0 = h0 = 0; the read-out and weight_hh see them through zero columns;
//   * the one place where the WIDTH itself enters the arithmetic - the layer-norm of the lambda gradients over the latent axis
//     (iodine.py:263-272, 376-384: mean and unbiased std over L) - runs over the real L (inner->Lreal, dz_latent_kernel), and so does the
//     logger's mean of init_mean / init_logvar.
// The outer handle owns the reference-shaped boundary: parameter table, padded copies of the parameters, the element maps, scratch for the
// tensors with a latent axis (eps, z, posterior), and the gradient in padded shape; every entry point forwards to the inner handle.
}  // namespace
struct PadShim {
    iodine_handle* inner = nullptr;
    int L = 0, H = 0, Lp = 0, Hp = 0;
    std::vector<float*> pparam;            // padded parameter copies [param]
    std::vector<int*> pmap;                // [param][padded element] -> element of the reference-shaped tensor, or -1 (zero)
    std::vector<int> pnumel;               // padded element counts
    float* pgrad = nullptr;                // flat gradient in padded shapes (the inner handle's named_parameters() order)
    std::vector<size_t> poff;
    size_t pgrad_total = 0;
    // scratch for one call's tensors with a latent axis: grown on demand (outside the refinement loop)
    size_t cap = 0;                        // floats per buffer
    float *eps = nullptr, *z = nullptr, *pm = nullptr, *plv = nullptr, *pm_in = nullptr, *plv_in = nullptr;
    std::vector<void*> owned;
};
namespace {

int shim_fail(iodine_handle* h, int rc)
{
    if (rc && h->shim && h->shim->inner) h->err = h->shim->inner->err.empty() ? std::string(iodine_last_error(nullptr)) : h->shim->inner->err;
    return rc;
}

// per-dimension index map of a concatenation of segments (real length -> padded length): padded index -> real index or -1
std::vector<int> seg_map(std::initializer_list<std::pair<int, int>> segs)
{
    std::vector<int> m;
    int real0 = 0;
    for (const auto& sg : segs) {
        for (int i = 0; i < sg.second; ++i) m.push_back(i < sg.first ? real0 + i : -1);
        real0 += sg.first;
    }
    return m;
}

int shim_build(iodine_handle* h)
{
    PadShim* sh = h->shim;
    const int L = sh->L, H = sh->H, Lp = sh->Lp, Hp = sh->Hp;
    iodine_handle* in = sh->inner;
    const size_t np = h->params.size();
    if (in->params.size() != np) return h->fail(IODINE_ERR_INVALID, "padded inner handle: parameter tables differ");
    sh->pparam.assign(np, nullptr); sh->pmap.assign(np, nullptr); sh->pnumel.assign(np, 0); sh->poff.assign(np, 0);
    auto ident = [](int n) { std::vector<int> m(n); for (int i = 0; i < n; ++i) m[i] = i; return m; };
    const std::vector<int> mH = seg_map({{H, Hp}}), mL = seg_map({{L, Lp}}), m4H = seg_map({{H, Hp}, {H, Hp}, {H, Hp}, {H, Hp}}),
                           mIN = seg_map({{H, Hp}, {L, Lp}, {L, Lp}, {L, Lp}, {L, Lp}}), mL2 = seg_map({{L, Lp}, {2, 2}});
    size_t off = 0;
    for (size_t i = 0; i < np; ++i) {
        const ParamInfo &pr = h->params[i], &pp = in->params[i];
        if (pr.name != pp.name || pr.ndim != pp.ndim) return h->fail(IODINE_ERR_INVALID, "padded inner handle: parameter " + pr.name + " differs");
        std::vector<int> dm[4];
        for (int d = 0; d < 4; ++d) dm[d] = ident((int)pp.dims[d]);
        const std::string& n = pr.name;
        if (n == "refine.mlp.layers.0.weight" || n == "refine.mlp.layers.0.bias") dm[0] = mH;
        else if (n == "refine.lstm.weight_ih") { dm[0] = m4H; dm[1] = mIN; }
        else if (n == "refine.lstm.weight_hh") { dm[0] = m4H; dm[1] = mH; }
        else if (n == "refine.lstm.bias_ih" || n == "refine.lstm.bias_hh") dm[0] = m4H;
        else if (n == "refine.mean_update.weight" || n == "refine.logvar_update.weight") { dm[0] = mL; dm[1] = mH; }
        else if (n == "refine.mean_update.bias" || n == "refine.logvar_update.bias" || n == "posterior.init_mean" || n == "posterior.init_logvar") dm[0] = mL;
        else if (n == "decoder.mlc.layers.0.weight") dm[1] = mL2;          // [Cd][L latent channels | x, y][k][k]
        for (int d = 0; d < 4; ++d)
            if ((long long)dm[d].size() != pp.dims[d]) return h->fail(IODINE_ERR_INVALID, "padded inner handle: shape of " + n);
        const size_t numel = pp.numel();
        std::vector<int> map(numel);
        size_t e = 0;
        for (int a = 0; a < (int)pp.dims[0]; ++a)
            for (int b2 = 0; b2 < (int)pp.dims[1]; ++b2)
                for (int c = 0; c < (int)pp.dims[2]; ++c)
                    for (int d = 0; d < (int)pp.dims[3]; ++d, ++e) {
                        const int ia = dm[0][a], ib = dm[1][b2], ic = dm[2][c], id = dm[3][d];
                        map[e] = (ia < 0 || ib < 0 || ic < 0 || id < 0) ? -1
                                 : (int)((((size_t)ia * pr.dims[1] + ib) * pr.dims[2] + ic) * pr.dims[3] + id);
                    }
        void *dmap = nullptr, *dpar = nullptr;
        HIPCHK(h, hipMalloc(&dmap, numel * sizeof(int))); sh->owned.push_back(dmap);
        HIPCHK(h, hipMemcpy(dmap, map.data(), numel * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipMalloc(&dpar, numel * sizeof(float))); sh->owned.push_back(dpar);
        sh->pmap[i] = (int*)dmap; sh->pparam[i] = (float*)dpar; sh->pnumel[i] = (int)numel; sh->poff[i] = off;
        off += numel;
    }
    sh->pgrad_total = off;
    void* g = nullptr;
    HIPCHK(h, hipMalloc(&g, off * sizeof(float))); sh->owned.push_back(g);
    sh->pgrad = (float*)g;
    return IODINE_OK;
}

// scratch for (rows x Lp) tensors of a call; rows_eps = (T + 1) * N for the noise, N for the others
int shim_scratch(iodine_handle* h, size_t floats)
{
    PadShim* sh = h->shim;
    if (floats <= sh->cap) return IODINE_OK;
    float** bufs[6] = {&sh->eps, &sh->z, &sh->pm, &sh->plv, &sh->pm_in, &sh->plv_in};
    for (float** b : bufs) {
        if (*b) { HIPCHK(h, hipDeviceSynchronize()); HIPCHK(h, hipFree(*b)); *b = nullptr; }
        void* q = nullptr;
        HIPCHK(h, hipMalloc(&q, floats * sizeof(float)));
        *b = (float*)q;
    }
    sh->cap = floats;
    return IODINE_OK;
}

}  // namespace

extern "C" {

int iodine_abi_version(void) { return IODINE_ABI_VERSION; }

const char* iodine_last_error(const iodine_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int iodine_create(const iodine_config* cfg, iodine_handle** out)
{
    if (out) *out = nullptr;
    if (!cfg || !out) { g_create_error = "null argument"; return IODINE_ERR_INVALID; }
    const std::string why = validate(*cfg);
    if (!why.empty()) { g_create_error = why; return IODINE_ERR_INVALID; }
    iodine_handle* h = new iodine_handle();
    h->cfg = *cfg;
    if (cfg->dim_latent % 4 != 0 || cfg->ref_mlp_units % 4 != 0) {
        // boundary handle of a zero-padded inner handle (PadShim above): owns the reference-shaped parameter table and the maps only
        h->L = cfg->dim_latent; h->T = cfg->iters; h->K = cfg->slots; h->S = cfg->img_size; h->P = h->S * h->S; h->H = cfg->ref_mlp_units;
        h->Cd = cfg->dec_conv_chan; h->Dd = cfg->dec_conv_layers; h->Cr = cfg->ref_conv_chan; h->Dr = cfg->ref_conv_layers;
        h->kd = cfg->dec_kernel_size; h->kr = cfg->ref_kernel_size; h->rs = cfg->ref_stride;
        h->n_in = 0;
        for (unsigned bit = 0; bit < 12; ++bit) {
            static const int cnt[12] = {0, 0, 3, 3, 1, 1, 1, 3, 1, 1, 1, 2};      // channels per ENCODING entry, order of the IODINE_ENC_* bits
            if (cfg->encoding & (1u << bit)) h->n_in += cnt[bit];
        }
        build_param_table(h);
        h->shim = new PadShim();
        h->shim->L = cfg->dim_latent; h->shim->H = cfg->ref_mlp_units;
        h->shim->Lp = (cfg->dim_latent + 3) / 4 * 4; h->shim->Hp = (cfg->ref_mlp_units + 3) / 4 * 4;
        iodine_config pc = *cfg;
        pc.dim_latent = h->shim->Lp; pc.ref_mlp_units = h->shim->Hp;
        const int rc = iodine_create(&pc, &h->shim->inner);
        if (rc) { iodine_destroy(h); return rc; }                           // (g_create_error holds the inner message)
        h->shim->inner->Lreal = cfg->dim_latent;
        const int rb = shim_build(h);
        if (rb) { g_create_error = h->err; iodine_destroy(h); return rb; }
        *out = h;
        return IODINE_OK;
    }
    h->L = cfg->dim_latent; h->T = cfg->iters; h->K = cfg->slots; h->S = cfg->img_size; h->P = h->S * h->S;
    h->Cd = cfg->dec_conv_chan; h->Dd = cfg->dec_conv_layers; h->Cr = cfg->ref_conv_chan; h->Dr = cfg->ref_conv_layers;
    h->H = cfg->ref_mlp_units;
    h->kd = cfg->dec_kernel_size; h->kr = cfg->ref_kernel_size; h->rs = cfg->ref_stride;
    h->generic = h->kd != 3 || (h->Cd != 32 && h->Cd != 64) || h->S % 16 != 0;
    h->gen_ref = h->kr != 3 || (h->Cr != 32 && h->Cr != 64) || h->S % 16 != 0 || h->rs != 2;
    {
        // image-shaped entries in CODE order (iodine.py:277-340) with their channel counts
        static const struct { unsigned bit; int first, count; } ent[10] = {
            {IODINE_ENC_IMAGE, 0, 3}, {IODINE_ENC_MEANS, 3, 3}, {IODINE_ENC_MASK, 6, 1}, {IODINE_ENC_MASK_LOGITS, 7, 1},
            {IODINE_ENC_MASK_POSTERIOR, 8, 1}, {IODINE_ENC_GRAD_MEANS, 9, 3}, {IODINE_ENC_GRAD_MASK, 12, 1},
            {IODINE_ENC_LIKELIHOOD, 13, 1}, {IODINE_ENC_LEAVE_ONE_OUT, 14, 1}, {IODINE_ENC_COORDINATE, 15, 2}};
        h->n_in = 0;
        for (const auto& e : ent)
            if (cfg->encoding & e.bit) for (int q = 0; q < e.count; ++q) h->enc_map[h->n_in++] = e.first + q;
        for (int j = h->n_in; j < 17; ++j) h->enc_map[j] = -1;
        h->enc_chmask = 0;
        for (int j = 0; j < h->n_in; ++j) h->enc_chmask |= 1u << h->enc_map[j];
    }
    build_param_table(h);

    auto bail = [&](hipError_t e, const char* what) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(e);
        iodine_destroy(h);
        return IODINE_ERR_HIP;
    };
#define ALLOC(ptr, n) do { hipError_t e_ = dev_alloc(h, &(ptr), (n)); if (e_ != hipSuccess) return bail(e_, "hipMalloc " #ptr); } while (0)
    const int L = h->L, Cd = h->Cd, Cr = h->Cr, H = h->H;
    ALLOC(h->lin, (size_t)h->S);
    ALLOC(h->wcls, (size_t)9 * L * Cd);
    ALLOC(h->wclsT, (size_t)9 * L * Cd);
    ALLOC(h->cmap, (size_t)h->P * Cd);
    h->dec_wf.assign(h->Dd, nullptr); h->dec_wb.assign(h->Dd, nullptr); h->dec_b.assign(h->Dd, nullptr);
    h->dec_wf16.assign(h->Dd, nullptr); h->dec_wb16.assign(h->Dd, nullptr); h->dec_wmeta.assign(h->Dd, nullptr);
    h->dec_wsf.assign(h->Dd, nullptr); h->dec_wsb.assign(h->Dd, nullptr);
    for (int l = 1; l < h->Dd; ++l) {
        ALLOC(h->dec_wsf[l], conv_ws_wpk_bytes(Cd) / 4);
        ALLOC(h->dec_wsb[l], conv_ws_wpk_bytes(Cd) / 4);
        ALLOC(h->dec_wf[l], conv_wpk_elems(Cd, Cd) * 4);
        ALLOC(h->dec_wb[l], conv_wpk_elems(Cd, Cd) * 4);
        ALLOC(h->dec_b[l], (size_t)Cd);
        ALLOC(h->dec_wf16[l], (size_t)(Cd / 16) * 9 * 2 * 2 * Cd * 4);      // fp16 x 8 per uint4 = 4 floats
        ALLOC(h->dec_wb16[l], (size_t)(Cd / 16) * 9 * 2 * 2 * Cd * 4);
        ALLOC(h->dec_wmeta[l], (size_t)4);
    }
    ALLOC(h->dec_out_w, (size_t)9 * Cd * 4);
    if (Cd == 64 || Cd == 32) ALLOC(h->dec_out_w32, (size_t)2 * (Cd / 2) * 64);
    ALLOC(h->dec_out_b, (size_t)4);
    ALLOC(h->dec_out_wb, conv_wpk_elems(4, Cd) * 4);
    ALLOC(h->dec_out_w16, (size_t)(Cd / 16) * 2 * 2 * 64 * 4);            // GEMM-form pack: [chunk][hi/lo][kh][64][8 fp16]
    ALLOC(h->dec_out_meta, (size_t)4);
    ALLOC(h->dec_out_wb16, (size_t)3 * 2 * 2 * Cd * 4);
    h->ref_w.assign(h->Dr, nullptr); h->ref_b.assign(h->Dr, nullptr);
    for (int l = 0; l < h->Dr; ++l) {
        ALLOC(h->ref_w[l], conv_wpk_elems(l == 0 ? 20 : Cr, Cr) * 4);
        ALLOC(h->ref_b[l], (size_t)Cr);
    }
    ALLOC(h->mlp_wT, (size_t)Cr * H); ALLOC(h->mlp_b, (size_t)H);
    ALLOC(h->wihT, (size_t)(H + 4 * L + H) * 4 * H);         // [W_ih^T ; W_hh^T] contiguous: one K-major operand for the gate GEMM (head_mfma)
    h->whhT = h->wihT + (size_t)(H + 4 * L) * 4 * H;
    ALLOC(h->lstm_b, (size_t)4 * H);
    ALLOC(h->wmT, (size_t)H * L); ALLOC(h->bm, (size_t)L); ALLOC(h->wvT, (size_t)H * L); ALLOC(h->bv, (size_t)L);
    ALLOC(h->init_mean, (size_t)L); ALLOC(h->init_logvar, (size_t)L);
    ALLOC(h->raw_mlp_w, (size_t)H * Cr); ALLOC(h->raw_wih, (size_t)4 * H * (H + 4 * L)); ALLOC(h->raw_whh, (size_t)4 * H * H);
    ALLOC(h->raw_wm, (size_t)L * H); ALLOC(h->raw_wv, (size_t)L * H);
    h->ref_wb.assign(h->Dr, nullptr);
    for (int l = 1; l < h->Dr; ++l) ALLOC(h->ref_wb[l], conv_wpk_elems(Cr, Cr) * 4);
    h->ref_wf16.assign(h->Dr, nullptr); h->ref_wb16.assign(h->Dr, nullptr); h->ref_wmeta.assign(h->Dr, nullptr);
    for (int l = 0; l < h->Dr; ++l) {
        ALLOC(h->ref_wf16[l], (size_t)((l == 0 ? 32 : Cr) / 16) * 9 * 2 * 2 * Cr * 4);
        ALLOC(h->ref_wmeta[l], (size_t)4);
        if (l > 0) ALLOC(h->ref_wb16[l], (size_t)(Cr / 16) * 9 * 2 * 2 * Cr * 4);
    }
    // split first layer: one 16-channel chunk each for the per-slot (12 real) and the per-image (8 real) channels
    ALLOC(h->ref_wk, (size_t)Cr * 12 * 9); ALLOC(h->ref_wsh, (size_t)Cr * 8 * 9); ALLOC(h->ref_g20, (size_t)Cr * 20 * 9);
    if (Cr % 32 == 0) { ALLOC(h->ref_w1ws, conv_ws_wpk_bytes(Cr) / 4); ALLOC(h->ref_w1ws_meta, (size_t)4); }
    h->ref_wsf.assign(h->Dr, nullptr); h->ref_wsf_meta.assign(h->Dr, nullptr);
    if (Cr == 64)
        for (int l = 1; l < h->Dr; ++l) { ALLOC(h->ref_wsf[l], conv_ws_wpk_bytes(Cr) / 4); ALLOC(h->ref_wsf_meta[l], (size_t)4); }
    ALLOC(h->ref_w17, (size_t)Cr * 17 * h->kr * h->kr); ALLOC(h->ref_g17, (size_t)Cr * 17 * h->kr * h->kr);
    if (h->generic) {
        const int kkd = h->kd * h->kd;
        h->gen_wdec.assign(h->Dd, nullptr);
        for (int l = 0; l < h->Dd; ++l) ALLOC(h->gen_wdec[l], (size_t)kkd * (l == 0 ? L + 2 : Cd) * Cd);
        ALLOC(h->gen_wout, (size_t)kkd * Cd * 4); ALLOC(h->gen_cterm, (size_t)h->P * Cd); ALLOC(h->gen_ident, (size_t)9 * Cd * L);
    }
    if (h->gen_ref) {
        const int kkr = h->kr * h->kr;
        h->gen_wref.assign(h->Dr, nullptr);
        for (int l = 0; l < h->Dr; ++l) ALLOC(h->gen_wref[l], (size_t)kkr * (l == 0 ? 17 : Cr) * Cr);
    }
    ALLOC(h->ref_wk16, (size_t)9 * 2 * 2 * Cr * 4); ALLOC(h->ref_wsh16, (size_t)9 * 2 * 2 * Cr * 4);
    ALLOC(h->ref_wkmeta, (size_t)4); ALLOC(h->ref_wshmeta, (size_t)4);
    {
        float* c_ = nullptr;
        ALLOC(c_, (size_t)4);
        h->elbo_counter = reinterpret_cast<unsigned*>(c_);
        if (hipError_t e_ = hipMemset(c_, 0, 16); e_ != hipSuccess) return bail(e_, "hipMemset elbo_counter");
    }
    if (Cr % 16 == 0) {
        float *pk = nullptr, *ps = nullptr;
        ALLOC(pk, refine_l0_wpk_bytes(Cr) / 4); ALLOC(ps, refine_l0_wpk_bytes(Cr) / 4);
        h->ref_l0k = pk; h->ref_l0s = ps;
        ALLOC(h->ref_l0kmeta, (size_t)4); ALLOC(h->ref_l0smeta, (size_t)4);
    }
    // gradient accumulators: ONE buffer, parameters back to back in named_parameters() order (the layout the wrapper's
    // flat gradient buffer has too), so that zeroing and the final scale-and-add are one launch each
    h->gacc.assign(h->params.size(), nullptr);
    h->gacc_total = 0;
    for (size_t i = 0; i < h->params.size(); ++i) h->gacc_total += h->params[i].numel();
    ALLOC(h->gacc_arena, h->gacc_total);
    {
        size_t off = 0;
        for (size_t i = 0; i < h->params.size(); ++i) { h->gacc[i] = h->gacc_arena + off; off += h->params[i].numel(); }
    }
#undef ALLOC
    std::vector<float> lin(h->S);
    iodine_linspace_host(h->S, lin.data());
    hipError_t e = hipMemcpy(h->lin, lin.data(), sizeof(float) * h->S, hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(e, "hipMemcpy linspace");
    *out = h;
    return IODINE_OK;
}

void iodine_destroy(iodine_handle* h)
{
    if (!h) return;
    if (h->shim) {
        if (h->shim->inner) iodine_destroy(h->shim->inner);
        for (void* p : h->shim->owned) (void)hipFree(p);
        for (float* p : {h->shim->eps, h->shim->z, h->shim->pm, h->shim->plv, h->shim->pm_in, h->shim->plv_in}) if (p) (void)hipFree(p);
        delete h->shim;
        h->shim = nullptr;
    }
    for (void* p : h->owned) (void)hipFree(p);
    for (auto& c : h->prof) for (auto& e : c.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (h->ws_own) (void)hipFree(h->ws_own);
    drop_graphs(h);
    delete h;
}

int iodine_num_params(const iodine_handle* h) { return h ? (int)h->params.size() : 0; }

int iodine_param_info(const iodine_handle* h, int index, const char** name, int* ndim, long long dims[4])
{
    if (!h || index < 0 || index >= (int)h->params.size()) return IODINE_ERR_INVALID;
    const ParamInfo& p = h->params[index];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = p.ndim;
    if (dims) for (int i = 0; i < 4; ++i) dims[i] = p.dims[i];
    return IODINE_OK;
}

int iodine_set_params(iodine_handle* h, void* stream, const float* const* dev, int n)
{
    if (!h) return IODINE_ERR_INVALID;
    if (n != (int)h->params.size() || !dev) return h->fail(IODINE_ERR_INVALID, "iodine_set_params: wrong parameter count");
    for (int i = 0; i < n; ++i)
        if (!dev[i]) return h->fail(IODINE_ERR_INVALID, "iodine_set_params: null pointer for " + h->params[i].name);
    hipStream_t st = (hipStream_t)stream;
    if (h->shim) {                                             // reference shapes -> zero-padded copies -> the inner handle
        PadShim* sh = h->shim;
        for (int i = 0; i < n; ++i) HIPCHK(h, launch_pad_gather(st, dev[i], sh->pmap[i], sh->pparam[i], sh->pnumel[i]));
        return shim_fail(h, iodine_set_params(sh->inner, stream, sh->pparam.data(), n));
    }
    auto P = [&](const std::string& name) { return dev[param_index(h, name)]; };
    // plain copies (biases, raw weights of the head backward) are collected and issued as one launch
    MultiCopy mc;
    mc.count = 0;
    auto queue_copy = [&](float* dst, const float* src, size_t count) -> hipError_t {
        if (mc.count == MCOPY_MAX) {
            const hipError_t e = launch_multi_copy(st, mc);
            if (e != hipSuccess) return e;
            mc.count = 0;
        }
        mc.src[mc.count] = src; mc.src2[mc.count] = nullptr; mc.dst[mc.count] = dst; mc.n[mc.count] = (int)count; mc.op[mc.count] = 0;
        mc.cols[mc.count] = 0; ++mc.count;
        return hipSuccess;
    };
    // (round 5) the head's transposes [R][Cc] -> [Cc][R] and the bias sum ride in the same launch as the plain copies
    auto queue_transpose = [&](float* dst, const float* src, int R, int Cc) -> hipError_t {
        const hipError_t e = queue_copy(dst, src, (size_t)R * Cc);
        if (e == hipSuccess) { mc.op[mc.count - 1] = 1; mc.cols[mc.count - 1] = Cc; }
        return e;
    };
    auto queue_add2 = [&](float* dst, const float* a, const float* b2, int count) -> hipError_t {
        const hipError_t e = queue_copy(dst, a, (size_t)count);
        if (e == hipSuccess) { mc.op[mc.count - 1] = 2; mc.src2[mc.count - 1] = b2; }
        return e;
    };
    const int L = h->L, Cd = h->Cd, Cr = h->Cr, H = h->H;
    // split-fp16 packs (power-of-two scale + layout per tensor and direction) are collected and issued as two launches at the end
    std::vector<PackJob> pj;
    auto pack_ws = [&](const float* src, int C, int tflip, float* meta, void* dst) {
        pj.push_back(PackJob{src, dst, meta, 0, {C, tflip, 0, 0, 0}});
        return hipSuccess;
    };
    auto pack_f16 = [&](const float* src, int O, int I, int cin, int cout, int tflip, float* meta, void* dst) {
        pj.push_back(PackJob{src, dst, meta, 1, {O, I, cin, cout, tflip}});
        return hipSuccess;
    };
    if (h->generic) {
        // fallback path: [tap][ci][co] packs of every conv, plain bias copies, the head below as on the tuned path
        for (int l = 0; l < h->Dd; ++l) {
            const std::string base = "decoder.mlc.layers." + std::to_string(l);
            HIPCHK(h, launch_gen_pack_weights(st, P(base + ".weight"), Cd, l == 0 ? L + 2 : Cd, h->kd, h->gen_wdec[l]));
            if (l > 0) HIPCHK(h, queue_copy(h->dec_b[l], P(base + ".bias"), Cd));
        }
        // bias + the conv of the two coordinate channels of the broadcast layer: the same map for every slot-image
        HIPCHK(h, launch_gen_l0_coord(st, h->gen_wdec[0], P("decoder.mlc.layers.0.bias"), h->lin, L, h->S, Cd, h->kd, h->gen_cterm));
        HIPCHK(h, launch_gen_pack_weights(st, P("decoder.conv.weight"), 4, Cd, h->kd, h->gen_wout));
        HIPCHK(h, queue_copy(h->dec_out_b, P("decoder.conv.bias"), 4));
        HIPCHK(h, launch_gen_identity(st, h->gen_ident, 9 * Cd, L));
    }
    if (h->gen_ref) {
        for (int l = 0; l < h->Dr; ++l) {
            const std::string base = "refine.mlc.layers." + std::to_string(l);
            const float* w = P(base + ".weight");
            if (l == 0 && h->n_in < 17) {
                HIPCHK(h, launch_enc_expand_weights(st, w, Cr, h->n_in, h->enc_map, h->ref_w17, h->kr * h->kr));
                w = h->ref_w17;
            }
            HIPCHK(h, launch_gen_pack_weights(st, w, Cr, l == 0 ? 17 : Cr, h->kr, h->gen_wref[l]));
            HIPCHK(h, queue_copy(h->ref_b[l], P(base + ".bias"), Cr));
        }
    }
    if (!h->generic) {
    // decoder
    HIPCHK(h, launch_dec_l0_prepare(st, P("decoder.mlc.layers.0.weight"), P("decoder.mlc.layers.0.bias"), h->lin, Cd, L,
                                    h->S, h->wcls, h->wclsT, h->cmap));
    for (int l = 1; l < h->Dd; ++l) {
        const float* w = P("decoder.mlc.layers." + std::to_string(l) + ".weight");
        if (conv_ws32_ok(h)) {                                 // exact-fp32 path, weight-stationary register layout (fp32)
            HIPCHK(h, launch_pack_conv_weights_ws32(st, w, Cd, 0, h->dec_wsf[l]));
            HIPCHK(h, launch_pack_conv_weights_ws32(st, w, Cd, 1, h->dec_wsb[l]));
        } else if (h->precision == 0) {                        // exact-fp32 path, LDS-tiled kernels (conv_precision invalidates the params)
            HIPCHK(h, launch_pack_conv_weights(st, w, Cd, Cd, Cd, Cd, 0, h->dec_wf[l]));
            HIPCHK(h, launch_pack_conv_weights(st, w, Cd, Cd, Cd, Cd, 1, h->dec_wb[l]));
        }
        // only the selected kernel's packs are maintained (a change of conv_variant / conv_precision invalidates the parameters)
        if (conv_ws_ok(h)) {                                   // weight-stationary register layout
            HIPCHK(h, pack_ws(w, Cd, 0, h->dec_wmeta[l], h->dec_wsf[l]));
            HIPCHK(h, pack_ws(w, Cd, 1, h->dec_wmeta[l] + 2, h->dec_wsb[l]));
        } else {
            HIPCHK(h, pack_f16(w, Cd, Cd, Cd, Cd, 0, h->dec_wmeta[l], h->dec_wf16[l]));
            HIPCHK(h, pack_f16(w, Cd, Cd, Cd, Cd, 1, h->dec_wmeta[l] + 2, h->dec_wb16[l]));
        }
        HIPCHK(h, queue_copy(h->dec_b[l], P("decoder.mlc.layers." + std::to_string(l) + ".bias"), Cd));
    }
    HIPCHK(h, queue_copy(h->dec_out_b, P("decoder.conv.bias"), 4));
    if (h->precision == 0) {                                   // exact-fp32 forms of the output conv (conv_precision invalidates the params)
        HIPCHK(h, launch_pack_dec_out(st, P("decoder.conv.weight"), h->dec_out_w, Cd));
        HIPCHK(h, launch_pack_conv_weights(st, P("decoder.conv.weight"), 4, Cd, 4, Cd, 1, h->dec_out_wb));
        if (h->dec_out_w32) HIPCHK(h, launch_pack_dec_out_rows32(st, P("decoder.conv.weight"), Cd, h->dec_out_w32));
    }
    // split-fp16 GEMM-form packs of the output conv, forward and data gradient (one scale): two more jobs of the batched pack
    pj.push_back(PackJob{P("decoder.conv.weight"), h->dec_out_w16, h->dec_out_meta, 3, {Cd, 0, 0, 0, 0}});
    pj.push_back(PackJob{P("decoder.conv.weight"), h->dec_out_wb16, h->dec_out_meta, 4, {Cd, 1, 0, 0, 0}});
    }   // !generic (decoder)
    if (!h->gen_ref) {
    // refinement conv stack
    const bool ref_fp32 = !refine_f16_ok(h);                    // the round-1 gather kernels: only where the tuned stride-2 kernels do not apply
    // first layer: the reference weight has n_in input channels (ARCH.ENCODING subset); the kernels see 17
    const float* w0 = P("refine.mlc.layers.0.weight");
    if (h->n_in < 17) {
        HIPCHK(h, launch_enc_expand_weights(st, w0, Cr, h->n_in, h->enc_map, h->ref_w17));
        w0 = h->ref_w17;
    }
    for (int l = 0; l < h->Dr; ++l) {
        const float* w = l == 0 ? w0 : P("refine.mlc.layers." + std::to_string(l) + ".weight");
        if (ref_fp32) HIPCHK(h, launch_pack_conv_weights(st, w, Cr, l == 0 ? 17 : Cr, l == 0 ? 20 : Cr, Cr, 0, h->ref_w[l]));
        HIPCHK(h, queue_copy(h->ref_b[l], P("refine.mlc.layers." + std::to_string(l) + ".bias"), Cr));
    }
    for (int l = 1; l < h->Dr && ref_fp32; ++l)
        HIPCHK(h, launch_pack_conv_weights(st, P("refine.mlc.layers." + std::to_string(l) + ".weight"), Cr, Cr, Cr, Cr, 2,
                                           h->ref_wb[l]));
    if (refine_f16_ok(h) && h->precision == 0) {
        // exact-fp32 path: fp32 weights in the same LDS-tile layouts (same buffers; conv_precision invalidates the parameters)
        for (int l = 0; l < h->Dr; ++l) {
            const float* w = l == 0 ? w0 : P("refine.mlc.layers." + std::to_string(l) + ".weight");
            HIPCHK(h, launch_pack_conv_weights_s2f32(st, w, Cr, l == 0 ? 17 : Cr, l == 0 ? 32 : Cr, Cr, 0, h->ref_wf16[l]));
            if (l > 0) HIPCHK(h, launch_pack_conv_weights_s2f32(st, w, Cr, Cr, Cr, Cr, 2, h->ref_wb16[l]));
        }
        HIPCHK(h, launch_ref_split_weights(st, w0, Cr, h->ref_wk, h->ref_wsh));
        HIPCHK(h, launch_pack_conv_weights_s2f32(st, h->ref_wk, Cr, 12, 16, Cr, 0, h->ref_wk16));
        HIPCHK(h, launch_pack_conv_weights_s2f32(st, h->ref_wsh, Cr, 8, 16, Cr, 0, h->ref_wsh16));
        if (Cr == 64)                                           // weight-stationary forward of layers 1 .. (fp32 weights in the same registers)
            for (int l = 1; l < h->Dr; ++l)
                HIPCHK(h, launch_pack_conv_weights_ws32(st, P("refine.mlc.layers." + std::to_string(l) + ".weight"), Cr, 0, h->ref_wsf[l]));
    } else if (refine_f16_ok(h)) {
        for (int l = 0; l < h->Dr; ++l) {
            const float* w = l == 0 ? w0 : P("refine.mlc.layers." + std::to_string(l) + ".weight");
            HIPCHK(h, pack_f16(w, Cr, l == 0 ? 17 : Cr, l == 0 ? 32 : Cr, Cr, 0, h->ref_wmeta[l],
                                                   h->ref_wf16[l]));
            if (l > 0)
                HIPCHK(h, pack_f16(w, Cr, Cr, Cr, Cr, 2, h->ref_wmeta[l] + 2, h->ref_wb16[l]));
        }
        // split first layer (refine_split): the same weights in the internal channel order, packed as two 16-channel convs
        HIPCHK(h, launch_ref_split_weights(st, w0, Cr, h->ref_wk, h->ref_wsh));
        HIPCHK(h, pack_f16(h->ref_wk, Cr, 12, 16, Cr, 0, h->ref_wkmeta, h->ref_wk16));
        HIPCHK(h, pack_f16(h->ref_wsh, Cr, 8, 16, Cr, 0, h->ref_wshmeta, h->ref_wsh16));
        if (h->ref_l0k) {                                      // fused encoding + layer 0: both parts as K = 16 MFMA operands
            pj.push_back(PackJob{h->ref_wk, h->ref_l0k, h->ref_l0kmeta, 2, {Cr, 12, 0, 0, 0}});
            pj.push_back(PackJob{h->ref_wsh, h->ref_l0s, h->ref_l0smeta, 2, {Cr, 8, 0, 0, 0}});
        }
        if (Cr == 64)                                           // weight-stationary forward of layers 1 ..
            for (int l = 1; l < h->Dr; ++l)
                HIPCHK(h, pack_ws(P("refine.mlc.layers." + std::to_string(l) + ".weight"), Cr, 0, h->ref_wsf_meta[l], h->ref_wsf[l]));
        if (h->Dr >= 2 && refine_bwd01_ok(h->S, Cr))           // fused layer-1 / layer-0 backward: W1 as the transposed conv's A operand
            HIPCHK(h, pack_ws(P("refine.mlc.layers.1.weight"), Cr, 1, h->ref_w1ws_meta, h->ref_w1ws));
    }
    }   // !gen_ref
    auto copy_raw = [&](float* dst, const std::string& name) {
        return queue_copy(dst, P(name), h->params[param_index(h, name)].numel());
    };
    HIPCHK(h, copy_raw(h->raw_mlp_w, "refine.mlp.layers.0.weight"));
    HIPCHK(h, copy_raw(h->raw_wih, "refine.lstm.weight_ih"));
    HIPCHK(h, copy_raw(h->raw_whh, "refine.lstm.weight_hh"));
    HIPCHK(h, copy_raw(h->raw_wm, "refine.mean_update.weight"));
    HIPCHK(h, copy_raw(h->raw_wv, "refine.logvar_update.weight"));
    // head
    HIPCHK(h, queue_transpose(h->mlp_wT, P("refine.mlp.layers.0.weight"), H, Cr));
    HIPCHK(h, queue_copy(h->mlp_b, P("refine.mlp.layers.0.bias"), H));
    HIPCHK(h, queue_transpose(h->wihT, P("refine.lstm.weight_ih"), 4 * H, H + 4 * L));
    HIPCHK(h, queue_transpose(h->whhT, P("refine.lstm.weight_hh"), 4 * H, H));
    HIPCHK(h, queue_add2(h->lstm_b, P("refine.lstm.bias_ih"), P("refine.lstm.bias_hh"), 4 * H));
    HIPCHK(h, queue_transpose(h->wmT, P("refine.mean_update.weight"), L, H));
    HIPCHK(h, queue_transpose(h->wvT, P("refine.logvar_update.weight"), L, H));
    HIPCHK(h, queue_copy(h->bm, P("refine.mean_update.bias"), L));
    HIPCHK(h, queue_copy(h->bv, P("refine.logvar_update.bias"), L));
    HIPCHK(h, queue_copy(h->init_mean, P("posterior.init_mean"), L));
    HIPCHK(h, queue_copy(h->init_logvar, P("posterior.init_logvar"), L));
    HIPCHK(h, launch_multi_copy(st, mc));
    if (!pj.empty()) HIPCHK(h, launch_pack_batch(st, pj.data(), (int)pj.size()));      // (behind ref_split / enc_expand: stream order)
    h->params_set = true;
    h->fwd_done = false;
    return IODINE_OK;
}

size_t iodine_workspace_bytes(const iodine_handle* h, int batch, int mode)
{
    if (!h || batch < 1) return 0;
    if (h->shim) return iodine_workspace_bytes(h->shim->inner, batch, mode);
    Arena q(nullptr); Buffers tmp; plan(h, batch, mode, q, tmp);
    return tmp.bytes;
}

int iodine_set_workspace(iodine_handle* h, void* dev_ptr, size_t bytes)
{
    if (!h) return IODINE_ERR_INVALID;
    if (((uintptr_t)dev_ptr & 255) != 0) return h->fail(IODINE_ERR_INVALID, "workspace must be 256-byte aligned");
    if (h->shim) return shim_fail(h, iodine_set_workspace(h->shim->inner, dev_ptr, bytes));
    h->ws_user = dev_ptr; h->ws_user_bytes = dev_ptr ? bytes : 0;
    h->buf = Buffers();
    h->fwd_done = false;
    h->last_elbo_iter = -1;
    return IODINE_OK;                    // graphs are keyed by the arena address (see ensure_workspace)
}

int iodine_set_option(iodine_handle* h, const char* key, double value)
{
    if (!h || !key) return IODINE_ERR_INVALID;
    if (h->shim) return shim_fail(h, iodine_set_option(h->shim->inner, key, value));
    if (!strcmp(key, "stop_after_iters")) { h->stop_after = (int)value; return IODINE_OK; }
    if (!strcmp(key, "profile")) { h->profile = (int)value; return IODINE_OK; }
    if (!strcmp(key, "graph")) { h->graph = value != 0; if (!h->graph) drop_graphs(h); return IODINE_OK; }
#ifdef IODINE_XSKIP_HOOK
    if (!strcmp(key, "xskip")) { g_iod_xskip = (int)value; return IODINE_OK; }       // timing-only ablation builds (common.h)
#endif
    if (!strcmp(key, "out_bwd_fused")) { h->out_bwd_fused = value != 0; return IODINE_OK; }
    if (!strcmp(key, "fuse_l0")) { h->fuse_l0 = value != 0; return IODINE_OK; }
    if (!strcmp(key, "refine_split")) { h->refine_split = value != 0; return IODINE_OK; }
    if (!strcmp(key, "head_fused")) { h->head_fused = value != 0; return IODINE_OK; }
    if (!strcmp(key, "refine_bwd_fused")) { h->refine_bwd_fused = value != 0; return IODINE_OK; }
    if (!strcmp(key, "profile_stride")) {
        if (value < 1) return h->fail(IODINE_ERR_INVALID, "profile_stride must be >= 1");
        h->profile_stride = (int)value; return IODINE_OK;
    }
    if (!strcmp(key, "head_mfma")) { h->head_mfma = value != 0; return IODINE_OK; }
    if (!strcmp(key, "refine_l0_fused")) { h->refine_l0_fused = value != 0; return IODINE_OK; }
    if (!strcmp(key, "refine_ws")) { h->refine_ws = value != 0; return IODINE_OK; }
    if (!strcmp(key, "dec_out_rows")) { h->dec_out_rows = value != 0; return IODINE_OK; }
    if (!strcmp(key, "wgrad_accum")) {
        if (h->wgrad_accum != (value != 0)) { h->buf = Buffers(); h->fwd_done = false; h->last_elbo_iter = -1; h->enc_valid = false; }   // the arena is re-planned
        h->wgrad_accum = value != 0; return IODINE_OK;
    }
    if (!strcmp(key, "conv_variant")) {
        if (value != 1 && value != 6) return h->fail(IODINE_ERR_INVALID, "conv_variant must be 1 (LDS-tiled) or 6 (weight-stationary)");
        if (((int)value == 6) != (h->variant == 6)) h->params_set = false;   // the other kernel's weight packs are not kept up to date
        h->variant = (int)value;
        return IODINE_OK;
    }
    if (!strcmp(key, "conv_precision")) {
        if (value != 0 && value != 1) return h->fail(IODINE_ERR_INVALID, "conv_precision must be 0 (f32) or 1 (f16x3)");
        if (h->precision != (int)value) h->params_set = false;    // the other path's weight packs are not kept up to date
        h->precision = (int)value;
        return IODINE_OK;
    }
    return h->fail(IODINE_ERR_INVALID, std::string("unknown option ") + key);
}

int iodine_reconstruct(iodine_handle* h, void* stream, int batch, const float* x, const float* eps, float* pred,
                       float* mask, float* mean, float* z, float* post_mean, float* post_logvar, float* elbo_iter)
{
    if (h && h->shim) {
        PadShim* sh = h->shim;
        if (batch < 1 || !x || !eps) return h->fail(IODINE_ERR_INVALID, "iodine_reconstruct: batch >= 1, x and eps are required");
        hipStream_t st = (hipStream_t)stream;
        const long long N = (long long)batch * h->K, R = (long long)(h->T + 1) * N;
        if (int r = shim_scratch(h, (size_t)R * sh->Lp)) return r;
        HIPCHK(h, launch_resize_rows(st, eps, sh->eps, R, sh->L, sh->Lp));
        const int rc = iodine_reconstruct(sh->inner, stream, batch, x, sh->eps, pred, mask, mean, z ? sh->z : nullptr, post_mean ? sh->pm : nullptr,
                                          post_logvar ? sh->plv : nullptr, elbo_iter);
        if (rc) return shim_fail(h, rc);
        if (z) HIPCHK(h, launch_resize_rows(st, sh->z, z, N, sh->Lp, sh->L));
        if (post_mean) HIPCHK(h, launch_resize_rows(st, sh->pm, post_mean, N, sh->Lp, sh->L));
        if (post_logvar) HIPCHK(h, launch_resize_rows(st, sh->plv, post_logvar, N, sh->Lp, sh->L));
        return IODINE_OK;
    }
    int rc = check_ready(h, batch);
    if (rc) return rc;
    if (!x || !eps) return h->fail(IODINE_ERR_INVALID, "iodine_reconstruct: x and eps are required");
    rc = ensure_workspace(h, batch, 0);
    if (rc) return rc;
    h->fwd_done = false;                                   // the arena is re-used: a saved training forward is gone
    hipStream_t st = (hipStream_t)stream;
    const int B = batch, N = B * h->K, T = h->T;
    const bool partial = h->stop_after >= 0 && h->stop_after <= T;     // debug: stop before the final sample/decode
    const int n_it = partial ? h->stop_after : T;
    auto body = [&]() -> int {
        Buffers& b = h->buf;
        const size_t eps_stride = (size_t)N * h->L;
        HIPCHK(h, launch_x_to_nhwc4(st, x, b.x4, B, h->P));
        HIPCHK(h, launch_posterior_init(st, h->init_mean, h->init_logvar, b.pm, b.plv, b.h[0], b.c[0], N, h->L, h->H));
        for (int i = 0; i < n_it; ++i) {
            int r = elbo_and_gradients(h, st, B, eps + (size_t)i * eps_stride, i, true);
            if (r) return r;
            r = refine_step(h, st, B, i, false);
            if (r) return r;
        }
        if (!partial) {
            // z = posterior.sample(); decode(z)   (iodine.py:103,110).  The decoder writes into the (now free) gradient
            // buffer so that buf.dec_out keeps the outputs of the LAST elbo() call: the reference's self.mean / self.mask
            // and its logger entries are those (iodine.py:226-239), not the final decode.
            HIPCHK(h, launch_dec_v(st, b.pm, b.plv, eps + (size_t)T * eps_stride, nullptr, h->wcls, b.z[T], b.V, N, h->L, h->Cd));
            const int r = decoder_forward(h, st, N, b.z[T], b.g);
            if (r) return r;
            HIPCHK(h, launch_final_out(st, b.g, pred, mask, mean, nullptr, B, h->K, h->P));
            if (z) HIPCHK(h, hipMemcpyAsync(z, b.z[T], sizeof(float) * eps_stride, hipMemcpyDeviceToDevice, st));
        }
        if (post_mean) HIPCHK(h, hipMemcpyAsync(post_mean, b.pm, sizeof(float) * eps_stride, hipMemcpyDeviceToDevice, st));
        if (post_logvar) HIPCHK(h, hipMemcpyAsync(post_logvar, b.plv, sizeof(float) * eps_stride, hipMemcpyDeviceToDevice, st));
        if (elbo_iter && n_it > 0) HIPCHK(h, hipMemcpyAsync(elbo_iter, b.scal, sizeof(float) * 3 * n_it, hipMemcpyDeviceToDevice, st));
        return IODINE_OK;
    };
    rc = run_graphed(h, st, graph_key(h, 1, B, {x, eps, pred, mask, mean, z, post_mean, post_logvar, elbo_iter}), body);
    if (rc) return rc;
    h->last_elbo_iter = n_it > 0 ? n_it - 1 : -1;
    h->last_elbo_batch = B;
    // (host state, outside the graphed body) did this call leave the encoding in the workspace?  refine_step: l0f && !keep_enc skips it
    h->enc_valid = n_it > 0 && (h->stop_after >= 0 || !(refine_split_on(h) && h->precision == 1 && h->refine_l0_fused && refine_l0_fused_ok(h->S, h->Cr, h->K)));
    return IODINE_OK;
}

int iodine_decode(iodine_handle* h, void* stream, int batch, const float* z, float* pred, float* mask, float* mean)
{
    if (h && h->shim) {
        PadShim* sh = h->shim;
        if (batch < 1 || !z) return h->fail(IODINE_ERR_INVALID, "iodine_decode: batch >= 1 and z are required");
        const long long N = (long long)batch * h->K;
        if (int r = shim_scratch(h, (size_t)N * sh->Lp)) return r;
        HIPCHK(h, launch_resize_rows((hipStream_t)stream, z, sh->z, N, sh->L, sh->Lp));
        return shim_fail(h, iodine_decode(sh->inner, stream, batch, sh->z, pred, mask, mean));
    }
    int rc = check_ready(h, batch);
    if (rc) return rc;
    if (!z) return h->fail(IODINE_ERR_INVALID, "iodine_decode: z is required");
    rc = ensure_workspace(h, batch, 0);
    if (rc) return rc;
    h->fwd_done = false;
    hipStream_t st = (hipStream_t)stream;
    const int N = batch * h->K;
    auto body = [&]() -> int {
        Buffers& b = h->buf;
        HIPCHK(h, launch_dec_v(st, nullptr, nullptr, nullptr, z, h->wcls, nullptr, b.V, N, h->L, h->Cd));
        const int r = decoder_forward(h, st, N, z, b.g);      // not buf.dec_out: that belongs to the last elbo() call
        if (r) return r;
        HIPCHK(h, launch_final_out(st, b.g, pred, mask, mean, nullptr, batch, h->K, h->P));
        return IODINE_OK;
    };
    return run_graphed(h, st, graph_key(h, 2, batch, {z, pred, mask, mean}), body);
}

int iodine_elbo(iodine_handle* h, void* stream, int batch, const float* x, const float* post_mean, const float* post_logvar,
                const float* eps, float* terms)
{
    if (h && h->shim) {
        PadShim* sh = h->shim;
        if (batch < 1 || !x || !eps) return h->fail(IODINE_ERR_INVALID, "iodine_elbo: batch >= 1, x and eps are required");
        if ((post_mean == nullptr) != (post_logvar == nullptr))
            return h->fail(IODINE_ERR_INVALID, "iodine_elbo: pass both post_mean and post_logvar, or neither");
        hipStream_t st = (hipStream_t)stream;
        const long long N = (long long)batch * h->K;
        if (int r = shim_scratch(h, (size_t)N * sh->Lp)) return r;
        HIPCHK(h, launch_resize_rows(st, eps, sh->eps, N, sh->L, sh->Lp));
        if (post_mean) {
            HIPCHK(h, launch_resize_rows(st, post_mean, sh->pm_in, N, sh->L, sh->Lp));
            HIPCHK(h, launch_resize_rows(st, post_logvar, sh->plv_in, N, sh->L, sh->Lp));
        }
        return shim_fail(h, iodine_elbo(sh->inner, stream, batch, x, post_mean ? sh->pm_in : nullptr, post_mean ? sh->plv_in : nullptr, sh->eps, terms));
    }
    int rc = check_ready(h, batch);
    if (rc) return rc;
    if (!x || !eps) return h->fail(IODINE_ERR_INVALID, "iodine_elbo: x and eps are required");
    if ((post_mean == nullptr) != (post_logvar == nullptr))
        return h->fail(IODINE_ERR_INVALID, "iodine_elbo: pass both post_mean and post_logvar, or neither");
    rc = ensure_workspace(h, batch, 0);
    if (rc) return rc;
    h->fwd_done = false;
    hipStream_t st = (hipStream_t)stream;
    const int B = batch, N = B * h->K;
    auto body = [&]() -> int {
        Buffers& b = h->buf;
        HIPCHK(h, launch_x_to_nhwc4(st, x, b.x4, B, h->P));
        if (post_mean) {
            HIPCHK(h, hipMemcpyAsync(b.pm, post_mean, sizeof(float) * (size_t)N * h->L, hipMemcpyDeviceToDevice, st));
            HIPCHK(h, hipMemcpyAsync(b.plv, post_logvar, sizeof(float) * (size_t)N * h->L, hipMemcpyDeviceToDevice, st));
        } else {
            HIPCHK(h, launch_posterior_init(st, h->init_mean, h->init_logvar, b.pm, b.plv, b.h[0], b.c[0], N, h->L, h->H));
        }
        const int r = elbo_and_gradients(h, st, B, eps, 0, false);
        if (r) return r;
        if (terms) HIPCHK(h, hipMemcpyAsync(terms, b.scal, sizeof(float) * 3, hipMemcpyDeviceToDevice, st));
        return IODINE_OK;
    };
    rc = run_graphed(h, st, graph_key(h, 3, B, {x, post_mean, post_logvar, eps, terms}), body);
    if (rc) return rc;
    h->last_elbo_iter = 0;
    h->last_elbo_batch = B;
    return IODINE_OK;
}

int iodine_last_elbo_outputs(iodine_handle* h, void* stream, int count, float* z, float* mean, float* mask,
                             float* mask_logits, float* pred)
{
    if (!h) return IODINE_ERR_INVALID;
    if (h->shim) {
        PadShim* sh = h->shim;
        const long long N = (long long)std::max(count, 0) * h->K;
        if (z) if (int r = shim_scratch(h, (size_t)N * sh->Lp)) return r;
        const int rc = iodine_last_elbo_outputs(sh->inner, stream, count, z ? sh->z : nullptr, mean, mask, mask_logits, pred);
        if (rc) return shim_fail(h, rc);
        if (z) HIPCHK(h, launch_resize_rows((hipStream_t)stream, sh->z, z, N, sh->Lp, sh->L));
        return IODINE_OK;
    }
    if (h->last_elbo_iter < 0 || h->buf.bytes == 0)
        return h->fail(IODINE_ERR_STATE, "iodine_last_elbo_outputs: no elbo() has run on the current workspace");
    if (count < 1 || count > h->last_elbo_batch)
        return h->fail(IODINE_ERR_INVALID, "iodine_last_elbo_outputs: count must be in 1..batch of the last call");
    hipStream_t st = (hipStream_t)stream;
    Buffers& b = h->buf;
    if (mean || mask || mask_logits || pred)
        HIPCHK(h, launch_final_out(st, b.dec_out, pred, mask, mean, mask_logits, count, h->K, h->P));
    if (z)
        HIPCHK(h, hipMemcpyAsync(z, b.z[h->last_elbo_iter], sizeof(float) * (size_t)count * h->K * h->L,
                                 hipMemcpyDeviceToDevice, st));
    return IODINE_OK;
}

int iodine_last_posterior(iodine_handle* h, void* stream, int count, float* post_mean, float* post_logvar)
{
    if (!h) return IODINE_ERR_INVALID;
    if (h->shim) {
        PadShim* sh = h->shim;
        const long long N = (long long)std::max(count, 0) * h->K;
        if (int r = shim_scratch(h, (size_t)N * sh->Lp)) return r;
        const int rc = iodine_last_posterior(sh->inner, stream, count, post_mean ? sh->pm : nullptr, post_logvar ? sh->plv : nullptr);
        if (rc) return shim_fail(h, rc);
        if (post_mean) HIPCHK(h, launch_resize_rows((hipStream_t)stream, sh->pm, post_mean, N, sh->Lp, sh->L));
        if (post_logvar) HIPCHK(h, launch_resize_rows((hipStream_t)stream, sh->plv, post_logvar, N, sh->Lp, sh->L));
        return IODINE_OK;
    }
    if (h->last_elbo_iter < 0 || h->buf.bytes == 0)
        return h->fail(IODINE_ERR_STATE, "iodine_last_posterior: no refinement has run on the current workspace");
    if (count < 1 || count > h->last_elbo_batch)
        return h->fail(IODINE_ERR_INVALID, "iodine_last_posterior: count must be in 1..batch of the last call");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = sizeof(float) * (size_t)count * h->K * h->L;
    if (post_mean) HIPCHK(h, hipMemcpyAsync(post_mean, h->buf.pm, n, hipMemcpyDeviceToDevice, st));
    if (post_logvar) HIPCHK(h, hipMemcpyAsync(post_logvar, h->buf.plv, n, hipMemcpyDeviceToDevice, st));
    return IODINE_OK;
}

int iodine_train_forward(iodine_handle* h, void* stream, int batch, const float* x, const float* eps, float* loss,
                         float* elbo_iter)
{
    if (h && h->shim) {
        PadShim* sh = h->shim;
        if (batch < 1 || !x || !eps || !loss) return h->fail(IODINE_ERR_INVALID, "iodine_train_forward: batch >= 1, x, eps and loss are required");
        const long long R = (long long)(h->T + 1) * batch * h->K;
        if (int r = shim_scratch(h, (size_t)R * sh->Lp)) return r;
        HIPCHK(h, launch_resize_rows((hipStream_t)stream, eps, sh->eps, R, sh->L, sh->Lp));
        return shim_fail(h, iodine_train_forward(sh->inner, stream, batch, x, sh->eps, loss, elbo_iter));
    }
    int rc = check_ready(h, batch);
    if (rc) return rc;
    if (!x || !eps || !loss) return h->fail(IODINE_ERR_INVALID, "iodine_train_forward: x, eps and loss are required");
    // the backward pass runs the refinement conv stack over all T iterations as one batch of T * N slot-images
    if ((size_t)batch * h->K * h->T * h->P * 20 >= ((size_t)1 << 31) || (size_t)batch * h->K * h->T * (size_t)ref_out_size(h, h->S) * ref_out_size(h, h->S) * h->Cr >= ((size_t)1 << 31))
        return h->fail(IODINE_ERR_INVALID, "batch too large for one device in training: batch * slots * iters * pixels * 20 must stay below "
                                           "2^31 (shard the images over ranks, iodine_amd.parallel)");
    rc = ensure_workspace(h, batch, 1);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int B = batch, N = B * h->K, T = h->T, L = h->L;
    h->fwd_done = false;
    auto body = [&]() -> int {
        Buffers& b = h->buf;
        const size_t eps_stride = (size_t)N * L;
        HIPCHK(h, hipMemsetAsync(h->gacc_arena, 0, sizeof(float) * h->gacc_total, st));
        HIPCHK(h, launch_x_to_nhwc4(st, x, b.x4, B, h->P));
        HIPCHK(h, launch_posterior_init(st, h->init_mean, h->init_logvar, b.pm, b.plv, b.h[0], b.c[0], N, L, h->H));
        for (int i = 0; i <= T; ++i) {
            const float alpha = -((float)(i + 1) / (float)(T + 1)) / (float)B;      // d loss / d (B * ELBO_i)
            int r = elbo_and_gradients(h, st, B, eps + (size_t)i * eps_stride, i, true, alpha);
            if (r) return r;
            if (i == 0) {
                // lambda_0 = init_mean / init_logvar repeated over (B, K) (iodine.py:615-616): their gradient is the
                // column sum of d loss / d lambda_0; later lambdas are detached from it (iodine.py:642-643)
                HIPCHK(h, launch_colsum(st, b.g_pm[0], N, L, L, alpha, h->gacc[param_index(h, "posterior.init_mean")]));
                HIPCHK(h, launch_colsum(st, b.g_plv[0], N, L, L, alpha, h->gacc[param_index(h, "posterior.init_logvar")]));
            }
            if (i < T) {
                r = refine_step(h, st, B, i, true);
                if (r) return r;
            }
        }
        HIPCHK(h, launch_loss(st, b.scal, T + 1, loss));
        if (elbo_iter) HIPCHK(h, hipMemcpyAsync(elbo_iter, b.scal, sizeof(float) * 3 * (T + 1), hipMemcpyDeviceToDevice, st));
        return IODINE_OK;
    };
    rc = run_graphed(h, st, graph_key(h, 4, B, {x, eps, loss, elbo_iter}), body);
    if (rc) return rc;
    h->fwd_done = true;
    h->enc_valid = true;                                   // training keeps the encoding of every iteration (the backward reads it)
    h->fwd_batch = B;
    h->fwd_split = refine_split_on(h);     // layout of the saved refinement inputs (refine_split is part of the graph key)
    h->last_elbo_iter = T;
    h->last_elbo_batch = B;
    return IODINE_OK;
}

static int train_backward_impl(iodine_handle* h, void* stream, float grad_scale, const float* grad_scale_dev,
                               float* const* param_grads, int n, int accumulate)
{
    if (!h) return IODINE_ERR_INVALID;
    if (h->shim) {
        // the inner handle writes its (scaled) gradient in padded shapes; the real entries are scattered (or added) into the caller's tensors
        PadShim* sh = h->shim;
        if (n != (int)h->params.size() || !param_grads) return h->fail(IODINE_ERR_INVALID, "iodine_train_backward: wrong parameter count");
        std::vector<float*> ptrs(h->params.size());
        for (size_t p = 0; p < ptrs.size(); ++p) ptrs[p] = sh->pgrad + sh->poff[p];
        const int rc = train_backward_impl(sh->inner, stream, grad_scale, grad_scale_dev, ptrs.data(), n, 0);
        if (rc) return shim_fail(h, rc);
        for (size_t p = 0; p < ptrs.size(); ++p)
            if (param_grads[p]) HIPCHK(h, launch_pad_scatter((hipStream_t)stream, ptrs[p], sh->pmap[p], param_grads[p], sh->pnumel[p], accumulate));
        return IODINE_OK;
    }
    if (!h->fwd_done) return h->fail(IODINE_ERR_STATE, "iodine_train_backward: no iodine_train_forward to differentiate");
    if (n != (int)h->params.size() || !param_grads) return h->fail(IODINE_ERR_INVALID, "iodine_train_backward: wrong parameter count");
    if (h->buf.mode != 1 || h->buf.B != h->fwd_batch)
        return h->fail(IODINE_ERR_STATE, "iodine_train_backward: the training workspace of the forward pass was re-planned");
    hipStream_t st = (hipStream_t)stream;
    std::vector<const void*> kp;
    for (int i = 0; i < n; ++i) kp.push_back(param_grads[i]);
    std::vector<uintptr_t> key = graph_key(h, 5, h->fwd_batch, {});
    for (const void* q : kp) key.push_back((uintptr_t)q);
    uint32_t gs_bits; memcpy(&gs_bits, &grad_scale, 4);
    key.push_back(gs_bits);
    key.push_back((uintptr_t)grad_scale_dev);
    key.push_back((uintptr_t)accumulate);
    key.push_back((uintptr_t)h->fwd_split);                // the backward body reads the saved inputs in the forward's layout
    auto body = [&]() -> int {
    Buffers& b = h->buf;
    const int B = h->fwd_batch, N = B * h->K, T = h->T, L = h->L, H = h->H, Cr = h->Cr, IN = H + 4 * L;
    auto G = [&](const std::string& name) { return h->gacc[param_index(h, name)]; };
    if (h->head_fused && head_bptt_fits(L, H, Cr)) {
        // the whole BPTT recurrence of the head in one launch (rows are independent: a block walks i = T-1 .. 0 for its rows)
        PROF(h, st, "head_bwd", launch_head_bptt(st, b.g_pm[0], b.g_plv[0], b.gates[0], b.c[0], b.u[0], h->raw_wm, h->raw_wv, h->raw_whh,
                                                 h->raw_wih, h->raw_mlp_w, b.ddm, b.ddv, b.dgates, b.ds, b.dpooled, T, N, B, L, H, Cr));
    } else {
    int cf = 0;                                            // carry buffer flip
    for (int i = T - 1; i >= 0; --i) {
        // d loss / d delta_i = -w_{i+1}/B * d(B*ELBO_{i+1})/d lambda_{i+1}   (lambda_{i+1} = detach(lambda_i) + delta_i)
        const float alpha = -((float)(i + 2) / (float)(T + 1)) / (float)B;
        float *ddm = b.ddm + (size_t)i * N * L, *ddv = b.ddv + (size_t)i * N * L;
        float *dgates = b.dgates + (size_t)i * N * 4 * H, *ds = b.ds + (size_t)i * N * H;
        HIPCHK(h, launch_scale(st, b.g_pm[i + 1], alpha, ddm, N * L));
        HIPCHK(h, launch_scale(st, b.g_plv[i + 1], alpha, ddv, N * L));
        const float* c1 = b.c[i + 1];
        // read-out layers act on the cell state (iodine.py:488-492)
        HIPCHK(h, launch_sgemm(st, 0, 0, N, H, L, 1.f, ddm, L, h->raw_wm, H, 0.f, b.dc1, H));
        HIPCHK(h, launch_sgemm(st, 0, 0, N, H, L, 1.f, ddv, L, h->raw_wv, H, 1.f, b.dc1, H));
        const bool last = (i == T - 1);
        HIPCHK(h, launch_lstm_bwd_pointwise(st, b.gates[i], b.c[i], c1, b.dc1, last ? nullptr : b.carry_h[cf],
                                            last ? nullptr : b.carry_c[cf], dgates, b.carry_c[cf ^ 1], N, H));
        HIPCHK(h, launch_sgemm(st, 0, 0, N, H, 4 * H, 1.f, dgates, 4 * H, h->raw_whh, H, 0.f, b.carry_h[cf ^ 1], H));
        cf ^= 1;
        // MLP (double ELU) and average pool
        HIPCHK(h, launch_sgemm(st, 0, 0, N, H, 4 * H, 1.f, dgates, 4 * H, h->raw_wih, IN, 0.f, b.dxin, H));
        HIPCHK(h, launch_mlp_bwd_pointwise(st, b.dxin, H, b.u[i], ds, N, H));
        HIPCHK(h, launch_sgemm(st, 0, 0, N, Cr, H, 1.f, ds, H, h->raw_mlp_w, Cr, 0.f, b.dpooled + (size_t)i * N * Cr, Cr));
    }
    }
    {
        // Weight gradients of the head: sums over the iterations of X_i^T D_i = ONE GEMM per parameter over all T * N rows
        // (the per-iteration operands lie back to back: c[1..T], xin[0..T-1], h[0..T-1], pooled[0..T-1] and ddm / ddv / dgates / ds)
        const int R = T * N;
        HIPCHK(h, launch_sgemm(st, 1, 0, L, H, R, 1.f, b.ddm, L, b.c[1], H, 1.f, G("refine.mean_update.weight"), H));
        HIPCHK(h, launch_sgemm(st, 1, 0, L, H, R, 1.f, b.ddv, L, b.c[1], H, 1.f, G("refine.logvar_update.weight"), H));
        HIPCHK(h, launch_colsum(st, b.ddm, R, L, L, 1.f, G("refine.mean_update.bias")));
        HIPCHK(h, launch_colsum(st, b.ddv, R, L, L, 1.f, G("refine.logvar_update.bias")));
        HIPCHK(h, launch_sgemm(st, 1, 0, 4 * H, IN, R, 1.f, b.dgates, 4 * H, b.xin[0], IN, 1.f, G("refine.lstm.weight_ih"), IN));
        HIPCHK(h, launch_sgemm(st, 1, 0, 4 * H, H, R, 1.f, b.dgates, 4 * H, b.h[0], H, 1.f, G("refine.lstm.weight_hh"), H));
        HIPCHK(h, launch_colsum(st, b.dgates, R, 4 * H, 4 * H, 1.f, G("refine.lstm.bias_ih")));
        HIPCHK(h, launch_colsum(st, b.dgates, R, 4 * H, 4 * H, 1.f, G("refine.lstm.bias_hh")));
        HIPCHK(h, launch_sgemm(st, 1, 0, H, Cr, R, 1.f, b.ds, H, b.pooled[0], Cr, 1.f, G("refine.mlp.layers.0.weight"), Cr));
        HIPCHK(h, launch_colsum(st, b.ds, R, H, H, 1.f, G("refine.mlp.layers.0.bias")));
    }
    {
        // Conv stack of the refinement network, last layer first, for ALL iterations at once: its inputs are detached
        // (iodine.py:343), so the T passes only meet in the weight gradients - one batch of T * N slot-images per layer
        // (saved inputs / activations of the iterations lie back to back, plan()) instead of T launches over N each: the
        // 8 x 8 ... 64 x 64 layers fill the chip 5x better and 4 (T - 1) x 3 launches disappear.
        const int NT = T * N;
        std::vector<int> sz(h->Dr + 1);
        sz[0] = h->S;
        for (int l = 0; l < h->Dr; ++l) sz[l + 1] = ref_out_size(h, sz[l]);
        const int sl = sz[h->Dr];
        HIPCHK(h, launch_pool_bwd(st, b.dpooled, b.ract[0][h->Dr - 1], b.rdpre[h->Dr - 1], NT, sl * sl, Cr));
        // round 4: the data gradient of layer 1 and the weight / bias gradient of layer 0 in ONE launch - d(pre-activation 0), the
        // largest tensor of this backward (T * N x 64 x 64 x 64 floats at cfg3), is produced and consumed on chip
        const bool fuse01 = h->refine_bwd_fused && h->fwd_split && !h->gen_ref && h->precision == 1 && refine_f16_ok(h) && h->Dr >= 2 &&
                            refine_bwd01_ok(h->S, Cr);
        for (int l = h->Dr - 1; l >= 0; --l) {
            const float* in = l == 0 ? b.enc[0] : b.ract[0][l - 1];
            const int cip = l == 0 ? 20 : Cr, ireal = l == 0 ? 17 : Cr;
            int nparts = 0, cipad = 0;
            const std::string base = "refine.mlc.layers." + std::to_string(l);
            // the layer's weight-gradient destination: the accumulator itself, or (first layer of an ARCH.ENCODING subset) a
            // 17-channel scratch that is gathered into the n_in-channel accumulator afterwards
            const bool gather0 = l == 0 && h->n_in < 17;
            float* gw_dst = gather0 ? h->ref_g17 : G(base + ".weight");
            if (gather0) HIPCHK(h, hipMemsetAsync(h->ref_g17, 0, (size_t)Cr * 17 * h->kr * h->kr * sizeof(float), st));
            if (h->gen_ref) {
                PROF(h, st, "gen_conv", launch_gen_conv_wgrad(st, in, b.rdpre[l], b.gen_scr, NT, sz[l], ireal, cip, ireal, Cr, h->kr, h->rs, 1.f,
                                                              gw_dst, G(base + ".bias"), l == 0 ? h->enc_chmask : 0xffffffffu));
            } else if (l == 0 && fuse01) {
                int nb = 0;
                PROF(h, st, "refine_bwd01", launch_refine_bwd01(st, b.rdpre[1], h->ref_w1ws, h->ref_w1ws_meta, b.ract[0][0], b.enck[0], b.encs[0],
                                                                b.wg_part, b.wg_part_b, NT, sz[0], Cr, h->K, &nparts, &cipad, &nb));
                HIPCHK(h, hipMemsetAsync(h->ref_g20, 0, (size_t)Cr * 20 * 9 * sizeof(float), st));
                HIPCHK(h, launch_wgrad_reduce(st, b.wg_part, nparts, cipad, Cr, Cr, 20, 20, 1.f, h->ref_g20, b.wg_fold,
                                              b.wg_part_b, nb, G(base + ".bias")));
                HIPCHK(h, launch_ref_unsplit_grad(st, h->ref_g20, Cr, gw_dst));
            } else if (l == 0 && h->fwd_split) {
                // split first layer: 12 per-slot + 8 per-image channels from two tensors, gradient in the internal channel
                // order, then added to the reference layout
                int nb = 0;
                PROF(h, st, "refine_wgrad", launch_conv3x3_s2_wgrad_f16x3(st, b.enck[0], b.rdpre[0], b.wg_part, b.wg_part_b, NT, sz[0],
                                                                          20, Cr, &nparts, &cipad, &nb, b.encs[0], h->K, h->precision == 0));
                HIPCHK(h, hipMemsetAsync(h->ref_g20, 0, (size_t)Cr * 20 * 9 * sizeof(float), st));
                HIPCHK(h, launch_wgrad_reduce(st, b.wg_part, nparts, cipad, Cr, Cr, 20, 20, 1.f, h->ref_g20, b.wg_fold,
                                              b.wg_part_b, nb, G(base + ".bias")));
                HIPCHK(h, launch_ref_unsplit_grad(st, h->ref_g20, Cr, gw_dst));
            } else if (refine_f16_ok(h)) {
                int nb = 0;
                PROF(h, st, "refine_wgrad", launch_conv3x3_s2_wgrad_f16x3(st, in, b.rdpre[l], b.wg_part, b.wg_part_b, NT, sz[l],
                                                                          cip, Cr, &nparts, &cipad, &nb, nullptr, 0, h->precision == 0));
                HIPCHK(h, launch_wgrad_reduce(st, b.wg_part, nparts, cipad, Cr, Cr, ireal, ireal, 1.f, gw_dst, b.wg_fold,
                                              b.wg_part_b, nb, G(base + ".bias")));
            } else {
                PROF(h, st, "refine_wgrad", launch_conv3x3_wgrad_gather(st, in, b.rdpre[l], b.wg_part, NT, sz[l], sz[l], cip, Cr, 2,
                                                                        &nparts, &cipad));
                HIPCHK(h, launch_wgrad_reduce(st, b.wg_part, nparts, cipad, Cr, Cr, ireal, ireal, 1.f, gw_dst, b.wg_fold));
                PROF(h, st, "refine_bias_grad", launch_colsum_tall(st, b.rdpre[l], NT * sz[l + 1] * sz[l + 1], Cr, 1.f,
                                                                   G(base + ".bias"), b.wg_part_b, (size_t)512 * 64));
            }
            if (gather0) HIPCHK(h, launch_enc_gather_grad(st, h->ref_g17, Cr, h->n_in, h->enc_map, G(base + ".weight"), h->kr * h->kr));
            if (l > 0 && !(l == 1 && fuse01)) {
                if (h->gen_ref)
                    PROF(h, st, "gen_conv", launch_gen_conv_dgrad(st, b.rdpre[l], h->gen_wref[l], b.ract[0][l - 1], b.rdpre[l - 1], NT, sz[l],
                                                                  Cr, Cr, Cr, h->kr, h->rs));
                else if (refine_f16_ok(h))
                    PROF(h, st, "refine_dgrad", launch_conv3x3_s2_dgrad_f16x3(st, b.rdpre[l], h->ref_wb16[l], h->ref_wmeta[l] + 2,
                                                                              b.ract[0][l - 1], b.rdpre[l - 1], NT, sz[l], Cr, h->precision == 0));
                else
                    PROF(h, st, "refine_dgrad", launch_conv3x3_gather_dgrad(st, b.rdpre[l], h->ref_wb[l], b.ract[0][l - 1],
                                                                            b.rdpre[l - 1], NT, sz[l], sz[l], Cr, 2));
            }
        }
    }
    bool flat = true;                                          // caller's gradients back to back in the same order?
    for (size_t p = 0; p < h->params.size() && flat; ++p)
        flat = param_grads[p] && param_grads[p] == param_grads[0] + (h->gacc[p] - h->gacc_arena);
    if (flat) {
        HIPCHK(h, launch_axpy_dev(st, h->gacc_arena, grad_scale, grad_scale_dev, param_grads[0], (int)h->gacc_total, accumulate));
        return IODINE_OK;
    }
    for (size_t p = 0; p < h->params.size(); ++p) {
        if (!param_grads[p]) continue;
        HIPCHK(h, launch_axpy_dev(st, h->gacc[p], grad_scale, grad_scale_dev, param_grads[p], (int)h->params[p].numel(), accumulate));
    }
    return IODINE_OK;
    };
    const int rc = run_graphed(h, st, key, body);
    // like autograd without retain_graph: the saved forward is consumed (a second backward would add the BPTT terms to the
    // accumulators twice); iodine_train_forward must run again first
    h->fwd_done = false;
    return rc;
}

int iodine_train_backward(iodine_handle* h, void* stream, float grad_scale, float* const* param_grads, int n)
{
    return train_backward_impl(h, stream, grad_scale, nullptr, param_grads, n, 1);
}

int iodine_train_backward_flat(iodine_handle* h, void* stream, const float* grad_loss_dev, float* flat_grads, int accumulate)
{
    if (!h) return IODINE_ERR_INVALID;
    if (!flat_grads) return h->fail(IODINE_ERR_INVALID, "iodine_train_backward_flat: flat_grads is required");
    std::vector<float*> ptrs(h->params.size());
    size_t off = 0;                                            // parameters back to back in named_parameters() order (= the gacc layout)
    for (size_t p = 0; p < h->params.size(); ++p) { ptrs[p] = flat_grads + off; off += h->params[p].numel(); }
    return train_backward_impl(h, stream, 1.f, grad_loss_dev, ptrs.data(), (int)ptrs.size(), accumulate ? 1 : 0);
}

int iodine_logger_scalars(iodine_handle* h, void* stream, float* out2)
{
    if (!h || !out2) return IODINE_ERR_INVALID;
    if (h->shim) return shim_fail(h, iodine_logger_scalars(h->shim->inner, stream, out2));
    if (!h->params_set) return h->fail(IODINE_ERR_STATE, "iodine_set_params has not been called");
    HIPCHK(h, launch_mean2((hipStream_t)stream, h->init_mean, h->init_logvar, h->Lreal > 0 ? h->Lreal : h->L, out2));
    return IODINE_OK;
}

int iodine_debug_copy(iodine_handle* h, void* stream, const char* name, int iter, float* dst, size_t max_floats,
                      size_t* n_floats)
{
    if (!h || !name) return IODINE_ERR_INVALID;
    if (h->shim) return shim_fail(h, iodine_debug_copy(h->shim->inner, stream, name, iter, dst, max_floats, n_floats));   // (padded widths)
    if (h->buf.bytes == 0) return h->fail(IODINE_ERR_STATE, "iodine_debug_copy: no workspace yet");
    Buffers& b = h->buf;
    const size_t N = (size_t)b.B * h->K, P = h->P, L = h->L;
    if (iter < 0 || iter > h->T) return h->fail(IODINE_ERR_INVALID, "iodine_debug_copy: bad iteration index");
    const std::string s(name);
    const float* src = nullptr; size_t n = 0;
    if (s == "z") { src = b.z[iter]; n = N * L; }
    else if (s == "dec_out") { src = b.dec_out; n = N * P * 4; }
    else if (s == "g") { src = b.g; n = N * P * 4; }
    else if (s == "enc") {
        n = N * P * 20;
        // with refine_l0_fused the inference loop never writes the encoding (kernels_refl0.hip keeps it on chip) unless stop_after_iters asks
        if (!h->enc_valid)
            return h->fail(IODINE_ERR_STATE, "iodine_debug_copy: enc was not materialised by the last call (set stop_after_iters >= 0 or refine_l0_fused=0)");
        if (refine_split_on(h)) {                          // joined back into the reference's 17 (+3 pad) channel order
            if (n_floats) *n_floats = n;
            if (!dst) return IODINE_OK;
            if (n > max_floats) return h->fail(IODINE_ERR_INVALID, "iodine_debug_copy: destination too small");
            HIPCHK(h, launch_enc_join((hipStream_t)stream, b.enck[iter], b.encs[iter], dst, (int)N, h->K, (int)P));
            return IODINE_OK;
        }
        src = b.enc[iter];
    }
    else if (s == "latent") { src = b.latent[iter]; n = N * 4 * L; }
    else if (s == "g_pm") { src = b.g_pm[iter]; n = N * L; }
    else if (s == "g_plv") { src = b.g_plv[iter]; n = N * L; }
    else if (s == "lnstat") { src = b.lnstat; n = N * 8; }
    else if (s == "Rc") { src = b.Rc; n = N * 9 * h->Cd; }
    else if (s == "V") { src = b.V; n = N * 9 * h->Cd; }
    else if (s == "h") { src = b.h[iter + 1]; n = N * h->H; }
    else if (s == "c") { src = b.c[iter + 1]; n = N * h->H; }
    else if (s == "pm") { src = b.pm; n = N * L; }
    else if (s == "plv") { src = b.plv; n = N * L; }
    else if (s == "scal") { src = b.scal; n = (size_t)(h->T + 1) * 3; }
    else if (s == "img_terms") { src = b.img_terms; n = (size_t)(h->T + 1) * b.B * 2; }
    else if (s.rfind("act", 0) == 0) {
        const int l = atoi(s.c_str() + 3);
        if (l < 0 || l >= h->Dd) return h->fail(IODINE_ERR_INVALID, "bad decoder layer");
        src = b.act[l]; n = N * P * h->Cd;
    } else if (s.rfind("ract", 0) == 0) {
        const int l = atoi(s.c_str() + 4);
        if (l < 0 || l >= h->Dr) return h->fail(IODINE_ERR_INVALID, "bad refinement layer");
        int sz = h->S; for (int j = 0; j <= l; ++j) sz = ref_out_size(h, sz);
        src = b.ract[iter][l]; n = N * sz * sz * h->Cr;
    } else return h->fail(IODINE_ERR_INVALID, "iodine_debug_copy: unknown buffer " + s);
    if (n_floats) *n_floats = n;
    if (dst) {
        if (n > max_floats) return h->fail(IODINE_ERR_INVALID, "iodine_debug_copy: destination too small");
        HIPCHK(h, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return IODINE_OK;
}

int iodine_profile_read(iodine_handle* h, const char* category, double* total_ms, long long* launches, int reset)
{
    if (!h || !category) return IODINE_ERR_INVALID;
    if (h->shim) return shim_fail(h, iodine_profile_read(h->shim->inner, category, total_ms, launches, reset));
    double tot = 0.0; long long cnt = 0;
    if (!strcmp(category, "graph_captures") || !strcmp(category, "graph_replays")) {    // hipGraph bookkeeping (option "graph")
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = category[6] == 'c' ? h->graph_captures : h->graph_replays;
        if (reset) { if (category[6] == 'c') h->graph_captures = 0; else h->graph_replays = 0; }
        return IODINE_OK;
    }
    if (!strncmp(category, "seen:", 5)) {               // launches of the category since the last reset, bracketed or not (profile_stride)
        for (auto& c : h->prof)
            if (c.name == category + 5) { cnt = (long long)c.seen_win; if (reset) c.seen_win = 0; }
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = cnt;
        return IODINE_OK;
    }
    for (auto& c : h->prof) {
        if (c.name != category) continue;
        for (size_t i = 0; i < c.used; ++i) {
            HIPCHK(h, hipEventSynchronize(c.ev[i].second));
            float ms = 0.f;
            HIPCHK(h, hipEventElapsedTime(&ms, c.ev[i].first, c.ev[i].second));
            tot += ms; ++cnt;
        }
        if (reset) c.used = 0;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = cnt;
    return IODINE_OK;
}

int iodine_adam_step(void* stream, const long long* ptrs_dev, const long long* offsets_dev, int n_tensors, long long total,
                     double lr, double beta1, double beta2, double eps, double weight_decay, int step)
{
    if (!ptrs_dev || !offsets_dev || n_tensors < 1 || total < 1 || step < 1) { g_create_error = "iodine_adam_step: bad argument"; return IODINE_ERR_INVALID; }
    const hipError_t e = launch_adam_multi((hipStream_t)stream, ptrs_dev, offsets_dev, n_tensors, total, lr, beta1, beta2, eps,
                                           weight_decay, step);
    if (e != hipSuccess) { g_create_error = std::string("iodine_adam_step: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

int iodine_ari_table(void* stream, const float* mask, const unsigned char* gt, int batch, int slots, int n_gt, int pixels,
                     int* table)
{
    if (!mask || !gt || !table || batch < 1 || slots < 1 || n_gt < 1 || pixels < 1 || (size_t)n_gt * slots > 8192) {
        g_create_error = "iodine_ari_table: bad argument";
        return IODINE_ERR_INVALID;
    }
    const hipError_t e = launch_ari_table((hipStream_t)stream, mask, gt, batch, slots, n_gt, pixels, table);
    if (e != hipSuccess) { g_create_error = std::string("iodine_ari_table: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

int iodine_randn(void* stream, float* out, long long n, unsigned long long seed, unsigned long long stream_id)
{
    if (!out || n < 1) { g_create_error = "iodine_randn: bad argument"; return IODINE_ERR_INVALID; }
    const hipError_t e = launch_randn_philox((hipStream_t)stream, out, n, seed, stream_id);
    if (e != hipSuccess) { g_create_error = std::string("iodine_randn: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

void iodine_linspace_host(int n, float* out)
{
    // ATen's CPU linspace for float: step = (end - start) / (n - 1); first half counts up from start,
    // second half counts down from end (symmetric), all in fp32.
    const float start = -1.f, end = 1.f;
    if (n == 1) { out[0] = start; return; }
    const float step = (end - start) / (float)(n - 1);
    const int halfway = n / 2;
    for (int i = 0; i < n; ++i)
        out[i] = i < halfway ? start + step * (float)i : end - step * (float)(n - i - 1);
}

// ---- operator-level test entry points ---------------------------------------------------------
static std::string g_op_error;

int iodine_op_conv3x3(void* stream, int mode, const float* in, const float* w, const float* bias, const float* aux,
                      float* out, int n, int ih, int iw, int w_o, int w_i, int cin_pad, int cout, int stride, int epi,
                      int tflip)
{
    hipStream_t st = (hipStream_t)stream;
    float* wpk = nullptr;
    if (mode == 5 || mode == 6 || mode == 13 || mode == 14) {   // stride-2 forward (5) / data gradient (6), split-fp16; 13 / 14: their exact-fp32 forms; ih = fine size
        float* meta = nullptr;
        const bool fwd = mode == 5 || mode == 13;
        const int f32 = mode >= 13;
        const int cp = fwd ? (cin_pad == 20 ? 32 : (cin_pad == 12 || cin_pad == 8 ? 16 : cin_pad)) : cin_pad;   // floats per pixel -> packed chunks
        const size_t bytes = (size_t)(cp / 16) * 9 * 2 * 2 * cout * 16;
        if (hipMalloc((void**)&wpk, bytes + 64) != hipSuccess) return IODINE_ERR_HIP;
        meta = (float*)((char*)wpk + bytes);
        hipError_t e2 = f32 ? launch_pack_conv_weights_s2f32(st, w, w_o, w_i, cp, cout, fwd ? 0 : 2, wpk)
                            : launch_pack_conv_weights_f16(st, w, w_o, w_i, cp, cout, fwd ? 0 : 2, meta, wpk);
        if (e2 == hipSuccess)
            e2 = fwd ? launch_conv3x3_s2_f16x3(st, in, wpk, meta, bias, out, n, ih, cin_pad, cout, nullptr, 0, f32)
                     : launch_conv3x3_s2_dgrad_f16x3(st, in, wpk, meta, aux, out, n, ih, cout, f32);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
        (void)hipFree(wpk);
        if (e2 != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3(s2 f16x3): ") + hipGetErrorString(e2); return IODINE_ERR_HIP; }
        return IODINE_OK;
    }
    if (mode == 9 || mode == 10) {  // weight-stationary split-fp16 kernel (per-cell max side buffer from launch_cell_max)
        if (cin_pad != cout || w_o != cout || w_i != cout || ih != iw || ih % 16 != 0) { g_create_error = "iodine_op_conv3x3(ws): shape"; return IODINE_ERR_INVALID; }
        const size_t wb = conv_ws_wpk_bytes(cout), tf = conv_ws_tmax_floats(n, ih);
        char* buf = nullptr;
        if (hipMalloc((void**)&buf, wb + 64 + 2 * tf * sizeof(float)) != hipSuccess) return IODINE_ERR_HIP;
        float* meta = (float*)(buf + wb);
        float *tin = (float*)(buf + wb + 64), *tout = tin + tf;
        hipError_t e2 = launch_pack_conv_weights_ws(st, w, cout, tflip, meta, buf);
        if (e2 == hipSuccess) e2 = launch_cell_max(st, in, tin, n, ih, cout);
        if (e2 == hipSuccess) e2 = launch_conv3x3_ws_f16x3(st, in, buf, meta, bias, aux, out, tin, tout, n, ih, cout, epi, 0);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
        (void)hipFree(buf);
        if (e2 != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3(ws): ") + hipGetErrorString(e2); return IODINE_ERR_HIP; }
        return IODINE_OK;
    }
    if (mode == 15 || mode == 16) { // weight-stationary STRIDE-2 conv c -> c + bias + ELU (kernels_refws.hip): split-fp16 (15) / exact fp32 (16); ih = fine size
        if (cin_pad != cout || w_o != cout || w_i != cout || ih != iw || !conv3x3_s2ws_ok(ih, cout)) { g_create_error = "iodine_op_conv3x3(s2ws): shape"; return IODINE_ERR_INVALID; }
        char* buf = nullptr;
        const size_t wb = conv_ws_wpk_bytes(cout);
        if (hipMalloc((void**)&buf, wb + 64) != hipSuccess) return IODINE_ERR_HIP;
        float* meta = (float*)(buf + wb);
        hipError_t e2 = mode == 16 ? launch_pack_conv_weights_ws32(st, w, cout, 0, buf) : launch_pack_conv_weights_ws(st, w, cout, 0, meta, buf);
        if (e2 == hipSuccess) e2 = launch_conv3x3_s2ws_f16x3(st, in, buf, meta, bias, out, n, ih, cout, mode == 16);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
        (void)hipFree(buf);
        if (e2 != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3(s2ws): ") + hipGetErrorString(e2); return IODINE_ERR_HIP; }
        return IODINE_OK;
    }
    if (mode == 12) {               // exact-fp32 form of the weight-stationary kernel (v_mfma_f32_16x16x4_f32)
        if (cin_pad != cout || w_o != cout || w_i != cout || ih != iw || ih % 16 != 0) { g_create_error = "iodine_op_conv3x3(ws f32): shape"; return IODINE_ERR_INVALID; }
        char* buf = nullptr;
        if (hipMalloc((void**)&buf, conv_ws_wpk_bytes(cout)) != hipSuccess) return IODINE_ERR_HIP;
        hipError_t e2 = launch_pack_conv_weights_ws32(st, w, cout, tflip, buf);
        if (e2 == hipSuccess) e2 = launch_conv3x3_ws_f32(st, in, buf, bias, aux, out, n, ih, cout, epi, 0);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
        (void)hipFree(buf);
        if (e2 != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3(ws f32): ") + hipGetErrorString(e2); return IODINE_ERR_HIP; }
        return IODINE_OK;
    }
#ifdef IODINE_WITH_WINO
    if (mode == 11) {               // Winograd F(2x2, 3x3) split-fp16 kernel (C = 64): experiment libraries only (tools/wino_variants.sh)
        if (cin_pad != cout || w_o != cout || w_i != cout || ih != iw || ih % 16 != 0) { g_create_error = "iodine_op_conv3x3(wino): shape"; return IODINE_ERR_INVALID; }
        const size_t wb = conv_wino_wpk_bytes(cout), tf = conv_ws_tmax_floats(n, ih), sf = conv_wino_scratch_floats(cout);
        char* buf = nullptr;
        if (hipMalloc((void**)&buf, wb + 64 + (2 * tf + sf) * sizeof(float)) != hipSuccess) return IODINE_ERR_HIP;
        float* meta = (float*)(buf + wb);
        float *tin = (float*)(buf + wb + 64), *tout = tin + tf, *scr = tout + tf;
        hipError_t e2 = launch_pack_conv_weights_wino(st, w, cout, tflip, meta, buf, scr);
        if (e2 == hipSuccess) e2 = launch_cell_max(st, in, tin, n, ih, cout);
        if (e2 == hipSuccess) e2 = launch_conv3x3_wino_f16x3(st, in, buf, meta, bias, aux, out, tin, tout, n, ih, cout, epi, 0);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
        (void)hipFree(buf);
        if (e2 != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3(wino): ") + hipGetErrorString(e2); return IODINE_ERR_HIP; }
        return IODINE_OK;
    }
#endif
    if (mode == 2) {                // split-fp16 LDS-tiled kernel
        float* meta = nullptr;
        const size_t bytes = (size_t)(cin_pad / 16) * 9 * 2 * 2 * cout * 16;
        if (hipMalloc((void**)&wpk, bytes + 64) != hipSuccess) return IODINE_ERR_HIP;
        meta = (float*)((char*)wpk + bytes);
        hipError_t e2 = launch_pack_conv_weights_f16(st, w, w_o, w_i, cin_pad, cout, tflip, meta, wpk);
        if (e2 == hipSuccess)
            e2 = launch_conv3x3_tile_f16x3(st, in, wpk, meta, bias, aux, out, n, ih, cin_pad, cout, epi, 0);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
        (void)hipFree(wpk);
        if (e2 != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3(f16x3): ") + hipGetErrorString(e2); return IODINE_ERR_HIP; }
        return IODINE_OK;
    }
    if (hipMalloc((void**)&wpk, conv_wpk_elems(cin_pad, cout) * 16) != hipSuccess) return IODINE_ERR_HIP;
    hipError_t e = launch_pack_conv_weights(st, w, w_o, w_i, cin_pad, cout, tflip, wpk);
    if (e == hipSuccess) {
        if (mode == 0) e = (ih == iw && stride == 1) ? launch_conv3x3_tile(st, in, wpk, bias, aux, out, n, ih, cin_pad, cout, epi)
                                                     : hipErrorInvalidValue;
        else e = launch_conv3x3_gather(st, in, wpk, bias, out, n, ih, iw, cin_pad, cout, stride);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(wpk);
    if (e != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

int iodine_op_dec_out(void* stream, const float* in, const float* w, const float* bias, float* out, int n, int s, int c)
{
    hipStream_t st = (hipStream_t)stream;
    float* wk = nullptr;
    if (hipMalloc((void**)&wk, (size_t)9 * c * 4 * sizeof(float)) != hipSuccess) return IODINE_ERR_HIP;
    hipError_t e = launch_pack_dec_out(st, w, wk, c);
    if (e == hipSuccess) e = launch_dec_out(st, in, wk, bias, out, n, s, c);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(wk);
    if (e != hipSuccess) { g_create_error = std::string("iodine_op_dec_out: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

int iodine_op_dec_out_f16x3(void* stream, const float* in, const float* w, const float* bias, float* out, int n, int s, int c, int variant)
{
    hipStream_t st = (hipStream_t)stream;
    const size_t wb = (size_t)(c / 16) * 2 * 2 * 64 * 16, tf = conv_ws_tmax_floats(n, s);
    char* buf = nullptr;
    if (s % 16 != 0 || (c != 64 && c != 32)) { g_create_error = "iodine_op_dec_out_f16x3: shape"; return IODINE_ERR_INVALID; }
    if (hipMalloc((void**)&buf, wb + 64 + tf * sizeof(float)) != hipSuccess) return IODINE_ERR_HIP;
    float* meta = (float*)(buf + wb);
    float* tin = (float*)(buf + wb + 64);
    hipError_t e = launch_pack_dec_out_gemm(st, w, c, meta, buf);
    if (e == hipSuccess) e = launch_cell_max(st, in, tin, n, s, c);
    if (e == hipSuccess && variant == 3) {          // exact-fp32 row-streaming form: its own weight operand (the buffer is large enough)
        e = launch_pack_dec_out_rows32(st, w, c, (float*)buf);
        if (e == hipSuccess) e = launch_dec_out_rows_f16x3(st, in, buf, nullptr, bias, out, n, s, c, nullptr, 1);
    } else if (e == hipSuccess)
        e = variant == 1 ? launch_dec_out_rows_f16x3(st, in, buf, meta, bias, out, n, s, c, tin)
                         : launch_dec_out_stream_f16x3(st, in, buf, meta, bias, out, n, s, c, variant == 2 ? nullptr : tin);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(buf);
    if (e != hipSuccess) { g_create_error = std::string("iodine_op_dec_out_f16x3: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

int iodine_op_conv3x3_wgrad(void* stream, const float* in, const float* d, float* gw, float* gb, int n, int s, int ci_pad,
                            int ci_real, int co, int stride)
{
    hipStream_t st = (hipStream_t)stream;
    const int cmax = std::max(std::max(ci_pad, co), 32);
    const size_t part_elems = (size_t)512 * 4 * 9 * cmax * cmax, fold_elems = (size_t)WGRAD_FOLD * 9 * cmax * cmax;
    float* buf = nullptr;
    if (hipMalloc((void**)&buf, (part_elems + fold_elems + (size_t)512 * 64) * sizeof(float)) != hipSuccess) return IODINE_ERR_HIP;
    float *part = buf, *fold = buf + part_elems, *part_b = fold + fold_elems;
    int nparts = 0, cip = 0, nb = 0;
    hipError_t e;
    if (stride == 1 && co == 4) {
        e = launch_dec_out_wgrad_gemm_f16x3(st, in, d, part, part_b, n, s, ci_pad, &nparts, &nb);
        if (e == hipSuccess) e = launch_wgrad_reduce(st, part, nparts, ci_pad, 4, 4, ci_real, ci_real, 1.f, gw, fold);
    } else if (stride == 1) {
        e = launch_conv3x3_wgrad_f16x3_ws(st, in, d, part, part_b, n, s, ci_pad, co, &nparts, &cip, &nb);
        // stride-1 partial tiles are [9][ci][co padded to 32]
        if (e == hipSuccess) e = launch_wgrad_reduce(st, part, nparts, ci_pad, cip, co, ci_real, ci_real, 1.f, gw, fold);
    } else {                                           // stride 2; stride -2: the exact-fp32 form of the same kernel
        e = launch_conv3x3_s2_wgrad_f16x3(st, in, d, part, part_b, n, s, ci_pad, co, &nparts, &cip, &nb, nullptr, 0, stride == -2);
        if (e == hipSuccess) e = launch_wgrad_reduce(st, part, nparts, cip, co, co, ci_real, ci_real, 1.f, gw, fold);
    }
    if (e == hipSuccess) e = launch_colsum(st, part_b, nb, co, co, 1.f, gb);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(buf);
    if (e != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3_wgrad: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

int iodine_op_gen_conv(void* stream, int mode, const float* in, const float* w, const float* bias, const float* aux, float* out, float* gb,
                       int n, int si, int ci, int ldc, int co, int k, int s, int elu)
{
    hipStream_t st = (hipStream_t)stream;
    if (mode < 0 || mode > 2 || (s != 1 && s != 2) || (k != 3 && k != 5 && k != 7) || ldc < ci) { g_create_error = "iodine_op_gen_conv: argument"; return IODINE_ERR_INVALID; }
    // tests only: elu bit 8 set = bits 9.. carry a per-channel mask of the input channels that can be non-zero (what the library hands
    // the stride-2 kernels for an ARCH.ENCODING subset: all-zero 4- / 16-channel groups are skipped) - the masked form must equal the plain one
    const unsigned chmask = (elu & 0x100) ? ((unsigned)elu >> 9) : 0xffffffffu;
    elu &= 1;
    float* buf = nullptr;
    // weights [tap][ci][co]: the forward pack has ci rows, the data-gradient pack ldc rows (the packed input-channel count is din's stride)
    const size_t wfl = (size_t)k * k * ldc * co, scr = mode == 2 ? gen_wgrad_scratch_floats(ci, co, k) : 0;
    if (hipMalloc((void**)&buf, (wfl + scr) * sizeof(float)) != hipSuccess) return IODINE_ERR_HIP;
    hipError_t e = hipSuccess;
    if (mode == 0) {
        e = launch_gen_pack_weights(st, w, co, ci, k, buf);
        if (e == hipSuccess) e = launch_gen_conv_fwd(st, in, buf, bias, out, n, si, ci, ldc, co, k, s, elu, chmask);
    } else if (mode == 1) {
        if (ci != ldc) { (void)hipFree(buf); g_create_error = "iodine_op_gen_conv: mode 1 needs ldc == ci"; return IODINE_ERR_INVALID; }
        e = launch_gen_pack_weights(st, w, co, ci, k, buf);
        if (e == hipSuccess) e = launch_gen_conv_dgrad(st, in, buf, aux, out, n, si, ci, ldc, co, k, s);
    } else {
        e = launch_gen_conv_wgrad(st, in, aux, buf + wfl, n, si, ci, ldc, ci, co, k, s, 1.f, out, gb, chmask);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(buf);
    if (e != hipSuccess) { g_create_error = std::string("iodine_op_gen_conv: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

int iodine_op_conv3x3_wgrad_f32(void* stream, const float* in, const float* d, float* gw, float* gb, int n, int s, int c)
{
    hipStream_t st = (hipStream_t)stream;
    const size_t part_elems = (size_t)512 * 4 * 9 * 32 * 32, fold_elems = (size_t)WGRAD_FOLD * 9 * 64 * 64;
    float* buf = nullptr;
    if (hipMalloc((void**)&buf, (part_elems + fold_elems + (size_t)512 * 64) * sizeof(float)) != hipSuccess) return IODINE_ERR_HIP;
    float *part = buf, *fold = buf + part_elems, *part_b = fold + fold_elems;
    int nparts = 0, cop = 0, nb = 0;
    hipError_t e;
    if (c < 0) {                                    // the output conv |c| -> 4 in GEMM form (d has 4 channels; gw [4][|c|][3][3], gb [4])
        c = -c;
        e = launch_dec_out_wgrad_f32(st, in, d, part, part_b, n, s, c, &nparts, &nb);
        if (e == hipSuccess) e = launch_wgrad_reduce(st, part, nparts, c, 4, 4, c, c, 1.f, gw, fold);
        if (e == hipSuccess) e = launch_colsum(st, part_b, nb, 4, 4, 1.f, gb);
    } else {
    e = launch_conv3x3_wgrad_f32_ws(st, in, d, part, part_b, n, s, c, &nparts, &cop, &nb);
    if (e == hipSuccess) e = launch_wgrad_reduce(st, part, nparts, c, cop, c, c, c, 1.f, gw, fold);
    if (e == hipSuccess) e = launch_colsum(st, part_b, nb, c, c, 1.f, gb);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(buf);
    if (e != hipSuccess) { g_create_error = std::string("iodine_op_conv3x3_wgrad_f32: ") + hipGetErrorString(e); return IODINE_ERR_HIP; }
    return IODINE_OK;
}

}  // extern "C"
