/* libiodine_hip.so -- C ABI of the MI355X-native (gfx950) IODINE refinement step.
 *
 * The reference (zhixuan-lin/IODINE) has no FFI: its boundary for this path is the
 * nn.Module protocol of lib/modeling/iodine.py as used by lib/engine/train.py:58-65,
 * lib/engine/eval.py:14-28 and lib/eval/ari_eval.py:22.  Each entry point below names the
 * reference interface it replaces.  All pointers named *_dev / x / eps / outputs are DEVICE
 * pointers owned by the caller (torch); `stream` is a hipStream_t (pass
 * torch.cuda.current_stream().cuda_stream).  Nothing throws across this ABI: every call
 * returns an int status and iodine_last_error() gives the message.
 *
 * Layouts at the boundary are the reference's: images (B,3,S,S) NCHW fp32 in [0,1];
 * eps (T+1,B,K,L) standard normals, one slice per Gaussian.sample call
 * (iodine.py:620-634); parameters in state_dict order with state_dict shapes (OIHW convs).
 * A handle is bound to one device and is NOT re-entrant (neither is the reference module,
 * iodine.py:36-52); use one handle per (device, stream).
 */
#ifndef IODINE_HIP_H
#define IODINE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IODINE_OK 0
#define IODINE_ERR_INVALID 1       /* bad argument / unsupported configuration */
#define IODINE_ERR_HIP 2           /* a HIP runtime call or kernel launch failed */
#define IODINE_ERR_STATE 3         /* call order violated (params not set, no forward before backward ...) */
#define IODINE_ERR_WORKSPACE 4     /* caller-provided workspace too small */

#define IODINE_ABI_VERSION 3

/* bit i set <=> the i-th entry of ARCH.ENCODING is enabled; order = code order of
 * IODINE.get_input_encoding (iodine.py:253-340).  Accepted: IODINE_ENC_FULL (every shipped IODINE config) and any list that keeps
 * 'posterior' and 'grad_post' and at least one image-shaped entry - in particular the reference's DEFAULT list
 * (lib/config/defaults.py:57-80: everything but 'coordinate', 15 input channels).  The first refinement layer's weight then has
 * that many input channels; absent channels are zero weights inside the library.  Lists without both latent entries are
 * rejected with a message, never approximated. */
#define IODINE_ENC_POSTERIOR      (1u << 0)
#define IODINE_ENC_GRAD_POST      (1u << 1)
#define IODINE_ENC_IMAGE          (1u << 2)
#define IODINE_ENC_MEANS          (1u << 3)
#define IODINE_ENC_MASK           (1u << 4)
#define IODINE_ENC_MASK_LOGITS    (1u << 5)
#define IODINE_ENC_MASK_POSTERIOR (1u << 6)
#define IODINE_ENC_GRAD_MEANS     (1u << 7)
#define IODINE_ENC_GRAD_MASK      (1u << 8)
#define IODINE_ENC_LIKELIHOOD     (1u << 9)
#define IODINE_ENC_LEAVE_ONE_OUT  (1u << 10)
#define IODINE_ENC_COORDINATE     (1u << 11)
#define IODINE_ENC_FULL           0xFFFu

/* Mirror of the ARCH.* node read by IODINE.__init__ (iodine.py:8-32; defaults lib/config/defaults.py:35-100). */
typedef struct iodine_config {
    int dim_latent;        /* ARCH.DIM_LATENT  (2..256; widths that are not multiples of 4 run zero-padded inside, same shapes at this boundary) */
    int iters;             /* ARCH.ITERS       */
    int slots;             /* ARCH.SLOTS       (1..16) */
    int img_size;          /* ARCH.IMG_SIZE    (multiples of 16: tuned kernels; other sizes >= 8: generic fallback path) */
    int img_channels;      /* ARCH.IMG_CHANNELS (3) */
    double sigma;          /* ARCH.SIGMA       */
    int layernorm;         /* ARCH.LAYERNORM   */
    int stop_gradient;     /* ARCH.STOP_GRADIENT (stored, unused: iodine.py:21 has no caller) */
    unsigned encoding;     /* ARCH.ENCODING as IODINE_ENC_* bits (see above for what is accepted) */
    int ref_conv_chan;     /* ARCH.REF.CONV_CHAN   (32 or 64: tuned kernels; other divisors of 256: generic fallback path) */
    int ref_conv_layers;   /* ARCH.REF.CONV_LAYERS */
    int ref_mlp_units;     /* ARCH.REF.MLP_UNITS   (1..1024; not a multiple of 4: zero-padded inside, iodine_debug_copy then shows the padded widths) */
    int ref_kernel_size;   /* ARCH.REF.KERNEL_SIZE (3: tuned kernels; 5, 7: generic fallback path, kernels_generic.hip) */
    int ref_stride;        /* ARCH.REF.STRIDE      (2: tuned kernels; 1, 3 .. 8: the refinement stack on the generic path) */
    int dec_conv_chan;     /* ARCH.DEC.CONV_CHAN   (32 or 64: tuned kernels; other multiples of 4 in 8..256: generic path) */
    int dec_conv_layers;   /* ARCH.DEC.CONV_LAYERS (>= 1) */
    int dec_kernel_size;   /* ARCH.DEC.KERNEL_SIZE (3: tuned kernels; 5 - the reference's default - and 7: generic path) */
} iodine_config;

typedef struct iodine_handle iodine_handle;

int iodine_abi_version(void);

/* IODINE(ARCH) -- iodine.py:8-52.  Builds the parameter table and the packed-weight buffers on the
 * current HIP device.  On failure *out is NULL and iodine_last_error(NULL) holds the reason. */
int iodine_create(const iodine_config* cfg, iodine_handle** out);
void iodine_destroy(iodine_handle* h);
const char* iodine_last_error(const iodine_handle* h);

/* model.named_parameters() / state_dict() surface (lib/solver/build.py:10-14, lib/utils/checkpoint.py:43,68). */
int iodine_num_params(const iodine_handle* h);
int iodine_param_info(const iodine_handle* h, int index, const char** name, int* ndim, long long dims[4]);

/* load_state_dict / "parameters changed" notification: repacks every weight into the kernels' layouts
 * (NHWC/MFMA quads, border-class sums and coordinate map of the broadcast layer, transposed head weights).
 * dev_ptrs[i] is the i-th parameter (state_dict order, reference shapes, contiguous fp32). */
int iodine_set_params(iodine_handle* h, void* stream, const float* const* dev_ptrs, int n);

/* Workspace: mode 0 = inference (reconstruct/decode), 1 = training.  If no workspace is installed the
 * library allocates one itself on first use (never inside the refinement loop). */
size_t iodine_workspace_bytes(const iodine_handle* h, int batch, int mode);
int iodine_set_workspace(iodine_handle* h, void* dev_ptr, size_t bytes);

/* pred, mask, mean = model.reconstruct(x) -- iodine.py:107-112 (encode :73-105 + decode :59-71).
 * Outputs (any may be NULL): pred (B,3,S,S), mask (B,K,1,S,S), mean (B,K,3,S,S) NCHW; z (B,K,L) = the final
 * sample; post_mean / post_logvar (B,K,L) = lambda after T updates; elbo_iter (T,3) = {ELBO, KL, LL} of each
 * elbo() call, batch means exactly as iodine.py:193,220,223. */
int iodine_reconstruct(iodine_handle* h, void* stream, int batch, const float* x, const float* eps,
                       float* pred, float* mask, float* mean, float* z, float* post_mean, float* post_logvar,
                       float* elbo_iter);

/* pred, mask, mean = model.decode(z) -- iodine.py:59-71. */
int iodine_decode(iodine_handle* h, void* stream, int batch, const float* z, float* pred, float* mask, float* mean);

/* elbo = model.elbo(x) -- IODINE.elbo, iodine.py:161-241: ONE sample z = mu + exp(logvar/2) * eps from the given posterior
 * (post_mean / post_logvar (B,K,L); both NULL = the initial posterior of Gaussian.init_unit, iodine.py:607-618), decode,
 * mixture log-likelihood and KL.  eps (B,K,L); terms (3) = {ELBO, KL, LL} (device, may be NULL).  The tensors the reference
 * leaves on `self` (z, mean, mask, mask_logits) and the `pred` it hands to the logger are read with
 * iodine_last_elbo_outputs. */
int iodine_elbo(iodine_handle* h, void* stream, int batch, const float* x, const float* post_mean, const float* post_logvar,
                const float* eps, float* terms);

/* self.z / self.mean / self.mask / self.mask_logits and pred of the LAST elbo() call (iodine.py:171-187,225) -- the one made by
 * iodine_elbo, the last refinement iteration of iodine_reconstruct (NOT its final decode: the reference's logger shows the
 * last elbo() call, iodine.py:226-239) or the final elbo of iodine_train_forward -- for the first `count` images of that
 * call's batch: z (count,K,L), mean (count,K,3,S,S), mask and mask_logits (count,K,1,S,S), pred (count,3,S,S); any may be
 * NULL.  count = 1 is what the logger side channel needs. */
int iodine_last_elbo_outputs(iodine_handle* h, void* stream, int count, float* z, float* mean, float* mask,
                             float* mask_logits, float* pred);

/* self.posterior.mean / self.posterior.logvar as the reference's module holds them after the last call (Gaussian.update,
 * iodine.py:636-645, leaves lambda_T on the module after forward / encode; a following model.elbo(x) samples from it,
 * iodine.py:170): post_mean / post_logvar (count,K,L) of the first `count` images of the last refinement call; either may be
 * NULL. */
int iodine_last_posterior(iodine_handle* h, void* stream, int count, float* post_mean, float* post_logvar);

/* loss = model(x) -- IODINE.forward, iodine.py:115-158.  loss (1) and elbo_iter (T+1,3) are device outputs.
 * Keeps what iodine_train_backward needs in the workspace (the autograd graph of the reference). */
int iodine_train_forward(iodine_handle* h, void* stream, int batch, const float* x, const float* eps,
                         float* loss, float* elbo_iter);

/* loss.backward() -- lib/engine/train.py:63.  Accumulates grad_scale * d loss / d param INTO param_grads[i]
 * (+=, like autograd's .grad accumulation; zero_grad is the caller's job, train.py:62).  Consumes the saved forward (autograd
 * without retain_graph): a second call, or a call after any other compute entry point re-used the workspace, returns
 * IODINE_ERR_STATE. */
int iodine_train_backward(iodine_handle* h, void* stream, float grad_scale, float* const* param_grads, int n);

/* The same with the caller's gradients as ONE buffer (parameters back to back in iodine_param_info order -- the layout
 * iodine_amd.IODINE hands to autograd and all-reduces in place) and autograd's incoming d(out)/d(loss) read from DEVICE memory
 * (grad_loss_dev, one float; NULL = 1): flat = (accumulate ? flat : 0) + *grad_loss_dev * d loss / d params.  One launch at the
 * end; no host round trip, no separate zero-fill. */
int iodine_train_backward_flat(iodine_handle* h, void* stream, const float* grad_loss_dev, float* flat_grads, int accumulate);

/* logger.update(init_mean=posterior.init_mean.mean(), init_logvar=posterior.init_logvar.mean()) -- iodine.py:156-157:
 * out2 (2, device) = the two means of the parameters last handed to iodine_set_params. */
int iodine_logger_scalars(iodine_handle* h, void* stream, float* out2);

/* torch.randn_like of Gaussian.sample (iodine.py:632) without ATen: n standard normals from Philox4x32-10 + Box-Muller,
 * counter-based (element e depends only on (seed, stream_id, e)); the wrapper passes one stream_id per call. */
int iodine_randn(void* stream, float* out, long long n, unsigned long long seed, unsigned long long stream_id);

/* optimizer.step() -- lib/engine/train.py:65 with the Adam built by lib/solver/build.py:5-16.  One fused launch over all
 * tensors.  ptrs_dev[4*t + {0,1,2,3}] = device addresses of {param, grad, exp_avg, exp_avg_sq} of tensor t (as int64),
 * offsets_dev[t] = first flat element index of tensor t, total = sum of the element counts; step counts from 1.
 * Semantics of torch.optim.Adam (amsgrad=False, maximize=False, coupled weight decay). */
int iodine_adam_step(void* stream, const long long* ptrs_dev, const long long* offsets_dev, int n_tensors, long long total,
                     double lr, double beta1, double beta2, double eps, double weight_decay, int step);

/* ARI evaluation epilogue -- lib/eval/ari_eval.py:25-39 + lib/utils/ari.py:36-52: per-pixel argmax over the K slot masks and
 * the integer contingency table[b][i][k] = |gt_i AND (argmax == k)|.  mask (B,K,1,S,S) fp32 (device, as returned by
 * iodine_reconstruct), gt (B,G,S,S) uint8 0/1 (device, padded with empty masks), table (B,G,K) int32 (device, overwritten).
 * The ARI formula itself (lib/utils/ari.py:6-33, a handful of scalars per image) stays on the host. */
int iodine_ari_table(void* stream, const float* mask, const unsigned char* gt, int batch, int slots, int n_gt, int pixels,
                     int* table);

/* Options: "stop_after_iters" (debug: run only the first v refinement iterations of reconstruct, no final decode),
 * "graph" (1: replay the fixed-shape launch sequence of reconstruct / decode / elbo / train_forward / train_backward through a
 * hipGraph per distinct argument tuple -- first call eager, second captured, later ones one hipGraphLaunch; needs a non-default
 * stream; ignored while "profile" is on; 0 -- default),
 * "profile" (bracket kernel launches with HIP events on the launch stream: 1 = the dominant "conv_tile_*" launches only --
 * 54 of ~330 per training step, what bench.py keeps on inside its timed region; 2 = every category; 0 = off),
 * "conv_precision" (3x3 convs of the decoder and refinement stacks: 0 = exact fp32 MFMA -- IEEE fp32 products, fp32 accumulate,
 * the reference's arithmetic (nn.Conv2d fp32, iodine.py:583), and with it the per-pixel mixture terms (sigmoid, slot softmax,
 * responsibilities: iodine.py:185-216) on libm expf + IEEE division like ATen, where the default path uses v_exp_f32 / v_rcp_f32
 * on their bounded arguments (pixel_terms.h); 1 = fp32 operands split into fp16 hi+lo with one power-of-two
 * scale per 8 x 16 cell, 3 fp16 MFMAs, fp32 accumulate: products carry >= 22 bits relative to the CELL maximum (tile-relative,
 * not element-relative) -- default; only the selected path's weight packs are maintained, so a change must be followed by
 * iodine_set_params before the next compute call),
 * "conv_variant" (stride-1 conv C -> C of the decoder, either precision: 6 = weight-stationary persistent kernel, weights in
 * registers -- default for power-of-two image sizes; 1 = LDS-tiled kernel, 16x16 tiles, two blocks per CU -- the fallback for
 * other sizes; like conv_precision a change must be followed by iodine_set_params),
 * "fuse_l0" (1 -- default: the last decoder data gradient reduces its result to the broadcast layer's row sums in its epilogue
 * instead of storing it; 0 = store and reduce in a second kernel),
 * "out_bwd_fused" (1 -- default: in training the output conv's data gradient and weight / bias gradient come from ONE pass
 * over the saved activation; 0 = two kernels, as iodine_reconstruct's data gradient + the GEMM-form weight gradient),
 * "refine_split" (1 -- default on the split-fp16 path: the first refinement layer is computed as a per-slot conv over the 11
 * encoding channels that differ between the slots of an image plus a per-image conv over the 6 they share; 0 = one conv
 * over the 20-float encoding per slot; a change takes effect with the next forward),
 * "head_fused" (1 -- default: the back-propagation through time of the refinement head runs as ONE launch, a block per 8 slots
 * walking the T iterations; 0 = nine launches per iteration -- also the automatic fallback when MLP_UNITS is too large for the
 * fused kernel's LDS footprint),
 * "refine_bwd_fused" (1 -- default: training backward, data gradient of refinement layer 1 + weight / bias gradient of layer 0
 * in ONE launch, d(pre-activation 0) never stored (kernels_refbwd.hip; power-of-two image sizes >= 64, split first layer);
 * 0 = the two launches),
 * "refine_ws" (1 -- default: forward stride-2 convs of refinement layers 1.. on the weight-stationary kernel
 * (kernels_refws.hip); 0 = the LDS-tiled stride-2 kernel),
 * "refine_l0_fused" (1 -- default: the 17-channel encoding of get_input_encoding (iodine.py:243-343) and the first refinement
 * layer in ONE kernel (kernels_refl0.hip) -- in inference the encoding is never written, so iodine_debug_copy("enc") then
 * needs "stop_after_iters" >= 0 or this option 0; 0 = pixel_pass2 writes the encoding, two convs read it),
 * "head_mfma" (1 -- default: the LSTM gate pre-activations of the refinement head as one fp32-MFMA GEMM over all slots,
 * three launches; 0 = the one-launch head kernel),
 * "dec_out_rows" (1 -- default: the output conv's forward runs as a row-streaming kernel -- a block walks a strip of image
 * rows, every row of the [pixels x 36] product is computed once and kept in an LDS ring until the rows above and below are
 * there -- for image sizes 32 / 64 / 128 on the split path; 0 = 16 x 16 tiles, each recomputing its 18 x 18 halo),
 * "wgrad_accum" (0 -- default: the partial weight-gradient tiles of a decoder launch are reduced right behind it; 1 = every block
 * keeps its partial tile over the T + 1 decoder passes of a training step (adds alpha_i x pass i; alpha_i = the pass's loss
 * weight) and the fixed-order reduction runs once per layer and step, 3 + 1 instead of 18 + 6 reductions per cfg3 step --
 * measured equal in time, so not the default; a change re-plans the workspace),
 * "profile_stride" (n >= 1, default 1: at "profile" level 1 only every n-th launch of a category is bracketed with events --
 * a pair of event records idles the GPU for ~12 us; iodine_profile_read("seen:<category>") returns how many launches the
 * category had in all),
 * "xskip" (timing-only ablation libraries built with -DIODINE_XSKIP_HOOK; absent from the product build).
 * (The A/B-only selections of rounds 1-2 -- conv_variant 5, wgrad_ws, out_variant, out_dgrad_variant, zigzag -- were retired in
 * round 3; their kernels and measurements live under tools/experiments/ and DESIGN.md 4.3-4.5.) */
int iodine_set_option(iodine_handle* h, const char* key, double value);
/* Sum of event-measured durations (ms) and number of launches of one kernel category since the last reset:
 * "conv_tile_fwd", "conv_tile_dgrad", "conv_tile_wgrad", "dec_out", "dec_out_dgrad", "dec_out_wgrad", "dec_out_bwd", "dec_l0",
 * "l0_reduce",
 * "pixel_pass1", "pixel_pass2", "refine_l0" (first refinement layer), "refine_l0f" (encoding + first refinement layer in one
 * kernel, option refine_l0_fused), "refine_conv" (the others), "refine_head", "refine_wgrad", "refine_dgrad", "refine_bwd01"
 * (fused layer-1 data gradient + layer-0 weight gradient, option refine_bwd_fused), "refine_bias_grad", "head_bwd", "gen_conv"
 * (the convs of the generic path), "gen_l0" (its spatial-broadcast layer: prefix-table forward, tap-sum backward).  "seen:<category>" returns in *launches the number of launches of <category> since the
 * last reset, bracketed or not (option profile_stride).  Synchronises on the recorded events.  Two more names report
 * the hipGraph bookkeeping of option "graph" in *launches: "graph_captures" (graphs instantiated) and "graph_replays". */
int iodine_profile_read(iodine_handle* h, const char* category, double* total_ms, long long* launches, int reset);
/* Copy an internal buffer of the last call (name as listed in DESIGN.md "workspace") to dst (device). */
int iodine_debug_copy(iodine_handle* h, void* stream, const char* name, int iter, float* dst, size_t max_floats,
                      size_t* n_floats);

/* ---- operator-level entry points (used by tests/ to check each kernel against the oracle) ------------- */
/* torch.linspace(-1, 1, n) in fp32, bit-exact restatement of ATen's CPU kernel (iodine.py:334-335,526-527). HOST. */
void iodine_linspace_host(int n, float* out);
/* 3x3 conv, NHWC activations, OIHW weights; mode 0: stride-1 LDS-tiled fp32 MFMA (epi 0 bias+ELU, 1 multiply by
 * ELU'(aux), 2 none; transpose_flip=1 computes the data-gradient conv), mode 1: strided gather kernel (bias+ELU),
 * mode 2: stride-1 LDS-tiled split-fp16 (3 MFMA) variant of mode 0, modes 9 / 10: the weight-stationary split-fp16 kernel,
 * mode 12: its exact-fp32 form (weights as fp32 in the same registers, v_mfma_f32_16x16x4_f32; conv_precision 0),
 * modes 5 / 6: split-fp16 stride-2 forward / data gradient of the refinement stack, modes 13 / 14: their exact-fp32 forms
 * (v_mfma_f32_32x32x2_f32, conv_precision 0), modes 15 / 16: the weight-stationary stride-2 conv c -> c of refinement layers 1 ..
 * (c = 64; split-fp16 / exact fp32). */
int iodine_op_conv3x3(void* stream, int mode, const float* in_nhwc, const float* w_oihw, const float* bias,
                      const float* aux, float* out_nhwc, int n, int ih, int iw, int w_o, int w_i, int cin_pad,
                      int cout, int stride, int epi, int transpose_flip);
int iodine_op_dec_out(void* stream, const float* in_nhwc, const float* w_oihw, const float* bias, float* out_nhwc4,
                      int n, int s, int c);
/* the split-fp16 forms of the output conv C -> 4 (GEMM + 9-tap sum): variant 0 = 16 x 16 tiles with halo recompute
 * (dec_out_stream_f16x3_kernel, per-cell max side buffer), 1 = row-streaming kernel without halo recompute
 * (dec_out_rows_f16x3_kernel, s in {32, 64, 128}), 2 = the tiled kernel without a side buffer, 3 = the row-streaming kernel on
 * exact fp32 MFMA (conv_precision 0). */
int iodine_op_dec_out_f16x3(void* stream, const float* in_nhwc, const float* w_oihw, const float* bias, float* out_nhwc4,
                            int n, int s, int c, int variant);
/* weight + bias gradient of a 3x3 conv (kernel-level tests): in_nhwc [n][s][s][ci_pad] (ci_real of them meaningful),
 * d_nhwc [n][so][so][co] with so = s (stride 1) or s/2 (stride 2); gw_oihw [co][ci_real][3][3] and gb [co] are
 * ACCUMULATED into.  Split-fp16 kernels (stride 1: decoder stack, stride 2: refinement stack); stride -2 selects the
 * exact-fp32 form of the stride-2 kernel (conv_precision 0). */
int iodine_op_conv3x3_wgrad(void* stream, const float* in_nhwc, const float* d_nhwc, float* gw_oihw, float* gb, int n,
                            int s, int ci_pad, int ci_real, int co, int stride);

/* the same for the stride-1 conv c -> c (c = 32 / 64) on the exact-fp32 path (conv_precision 0): persistent, prefetched
 * v_mfma_f32_32x32x2_f32 kernel (kernels_wgrad32.hip); gw_oihw / gb are ACCUMULATED into.  c < 0: the output conv |c| -> 4 in
 * GEMM form (d_nhwc has 4 channels, gw_oihw [4][|c|][3][3], gb [4]). */
int iodine_op_conv3x3_wgrad_f32(void* stream, const float* in_nhwc, const float* d_nhwc, float* gw_oihw, float* gb, int n,
                                int s, int c);

/* the generic path's convolution (any odd kernel size k, any stride s in {1, 2}; kernels_generic.hip / kernels_gens2.hip), one
 * direction per call - mode 0: out_nhwc [n][so][so][co] = act(bias + conv(in_nhwc [n][si][si][ldc], w_oihw [co][ci][k][k])), elu = 1
 * applies ELU; mode 1: out_nhwc [n][si][si][ci] = ELU'(aux) * data gradient of the gradient in_nhwc [n][so][so][co]; mode 2: weight +
 * bias gradient of (in_nhwc, aux = gradient [n][so][so][co]) ADDED to out_nhwc = gw [co][ci][k][k] and gb [co].  so = (si - 1) / s + 1.
 * Modes 0 / 2, tests only: elu | 0x100 | (mask << 9) hands the kernels the per-channel mask of input channels that can be non-zero (bit c =
 * channel c; ci <= 22), as the library does for the first refinement layer of an ARCH.ENCODING subset - channel groups that are zero in
 * the input AND the weights are skipped; the result must equal the unmasked call. */
int iodine_op_gen_conv(void* stream, int mode, const float* in_nhwc, const float* w_oihw, const float* bias, const float* aux,
                       float* out_nhwc, float* gb, int n, int si, int ci, int ldc, int co, int k, int s, int elu);

#ifdef __cplusplus
}
#endif
#endif /* IODINE_HIP_H */
