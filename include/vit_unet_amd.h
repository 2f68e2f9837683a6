/* vit_unet_amd.h - C ABI of libvitunet_amd.so, the MI355X (gfx950) ViT-UNet forward/backward path.
 *
 * This is the drop-in boundary for the hot path of benayas1/vit-unet.  The
 * reference has no native code: every entry point below replaces a run of PyTorch ATen calls
 * made by /root/reference/vit_unet/torch/model.py (cited per function as model.py:LINES).  The
 * Python package vit-unet_amd/vit_unet binds these symbols with ctypes (INTEGRATION.md shows the
 * stub) and mirrors the reference's nn.Module surface on top of them.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless stated otherwise;
 *  - all work is enqueued on the caller's `stream` (a hipStream_t passed as void*); no hidden
 *    synchronisation, no allocation; state between calls: the thread-local last-error text, the
 *    opt-in launch profiler below (process-global, one instrumented stream at a time), and two
 *    opt-in experiment switches that keep per-thread streams / events (VU_FLASH_FORK=1,
 *    VU_SIDE_LANE=1: off by default).  Environment switches (VU_ATTN_FLASH, VU_TSGEMM, ...) are
 *    A/B measurement aids read at call time;
 *  - return 0 (VU_OK) on success, negative VU_E* otherwise; vu_last_error() gives the text;
 *  - `dtype` selects the STORAGE type of activations / GEMM weights: 0 = fp32, 1 = bf16.
 *    Arithmetic is always fp32 (fp32 MFMA accumulators).  Parameters that are not GEMM operands
 *    (conv taps, head-mix matrix, BatchNorm / LayerNorm affine, biases, positional embedding)
 *    are always read from the fp32 master arena.
 */
#ifndef VIT_UNET_AMD_H
#define VIT_UNET_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VU_OK 0
#define VU_EINVAL (-1)
#define VU_EUNSUPPORTED (-2)
#define VU_EWORKSPACE (-3)
#define VU_ELAUNCH (-4)

int vu_version(void);      /* 200 = this header; bumped whenever a struct or a signature changes */
int vu_config_size(void);  /* sizeof(vu_config) as the library was built: a binding compares it with its own struct before any call */
const char* vu_last_error(void);

/* HViT_UNet constructor arguments (model.py:264-299). */
typedef struct vu_config {
  int depth, depth_te, size_bottleneck;
  int im_size, patch_size, num_channels, hidden_dim, num_heads;
  float attn_drop, proj_drop, linear_drop;
  int out_conv; /* preprocessing == 'conv' (model.py:369-370, 427-428) */
  int dtype;    /* 0 fp32, 1 bf16 */
  int attn_operands; /* 0: q, k, v enter the attention products as stored; 1: rounded to OCP e4m3 first
                      * (BASELINE config 5, "fp8 attention operands"; see vu_round_e4m3) */
} vu_config;

/* One parameter of the flat arena, in the reference's registration order (model.py:309-370);
 * `name` is the reference state_dict key.  Offsets are in elements and 8-element aligned. */
typedef struct vu_param_entry {
  char name[96];
  long long offset;
  int ndim;
  int shape[4];
  int bn_index; /* >= 0 for var_norm.weight: index of this module's running stats */
} vu_param_entry;

int vu_model_validate(const vu_config* cfg);                    /* asserts of model.py:281-283 */
long long vu_model_param_elems(const vu_config* cfg);           /* arena length (padded)       */
int vu_model_num_params(const vu_config* cfg);
int vu_model_param_table(const vu_config* cfg, vu_param_entry* out, int capacity);
int vu_model_num_attn(const vu_config* cfg);                    /* BatchNorm modules           */
size_t vu_model_workspace_bytes(const vu_config* cfg, int B);   /* = vu_model_workspace_bytes_ex(cfg, B, 1): fits either mode */
/* The workspace of an eval-mode forward (training = 0; model.py:372-435 under model.eval(), functions.py:7-19) leaves out what only a
 * training step uses: the probability caches of the recompute attention (vu_set_flash_pcache).  vu_model_forward / _backward carve by
 * their own `training` argument, so a workspace sized for training serves both and one sized for eval is refused by a training call
 * (VU_EWORKSPACE).  vu_model_pcache_bytes: the part of the training workspace that is probability cache (after the budget below). */
size_t vu_model_workspace_bytes_ex(const vu_config* cfg, int B, int training);
size_t vu_model_pcache_bytes(const vu_config* cfg, int B);
/* Diagnostic (no reference counterpart): the carve of that workspace as text, one "name offset bytes" line per buffer, the
 * forward's buffers in execution order.  Returns the length of the full text (snprintf convention), < 0 on a bad config. */
int vu_model_workspace_describe(const vu_config* cfg, int B, char* out, int cap);
/* 1 when a training step of this configuration at B images per GPU is faster launched eagerly than replayed from a captured
 * graph: the recompute attention's backward then runs its dv sweep on a low-priority stream in the tails of the dq / dk
 * sweeps (more workgroups than the chip holds at once), and a captured kernel node carries no stream priority.  Needs a
 * current device.  (No reference counterpart: launch policy of this library.) */
int vu_model_prefers_eager(const vu_config* cfg, int B);

/* HViT_UNet.forward (model.py:372-435).
 *  params      fp32 master arena            shadow   bf16 copy of the arena (dtype 1) or NULL
 *  bn_state    fp32 [num_attn][2][heads] running_mean / running_var (updated when training)
 *  x, y        fp32 (B, C, im, im)          ws       workspace of vu_model_workspace_bytes()
 *  training    1: BatchNorm batch statistics + dropout (seed); 0: running statistics
 *  rng_salt    optional device uint32 mixed into the dropout keys (hipGraph replay), or NULL */
int vu_model_forward(const vu_config* cfg, const float* params, const void* shadow, float* bn_state,
                     const float* x, float* y, void* ws, size_t ws_bytes, int B, int training,
                     uint64_t seed, const uint32_t* rng_salt, void* stream);
/* Backward of the forward that last filled `ws`.  Parameter gradients are ACCUMULATED into the
 * fp32 arena `grads`; dx (fp32 (B,C,im,im)) may be NULL.  stage: 0 = all, 1 = output conv +
 * decoders + skips, 2 = bottleneck, 3 = encoders + positional embedding (for bucketed
 * all-reduce overlap; stages must be run in order 1,2,3). */
int vu_model_backward(const vu_config* cfg, const float* params, const void* shadow,
                      const float* bn_state, float* grads, const float* dy, float* dx, void* ws,
                      size_t ws_bytes, int B, int training, uint64_t seed, const uint32_t* rng_salt,
                      int stage, void* stream);

/* The same backward cut into UNITS in reverse execution order (unit 0 = output conv; then per decoder block, last
 * first, preceded by the SkipConnection that follows it in the forward; bottleneck blocks; encoder blocks; last unit =
 * positional embedding + dx), so that a data-parallel caller can start the all-reduce of a gradient bucket as soon
 * as the units producing it are enqueued.  vu_model_backward_unit_ranges fills lo_hi[2u], lo_hi[2u+1] with
 * the arena range [lo, hi) of unit u's parameter gradients.  Units must be run in order, each exactly once per step;
 * any split of [0, n) into calls is equivalent to vu_model_backward(stage = 0). */
int vu_model_num_backward_units(const vu_config* cfg);
int vu_model_backward_unit_ranges(const vu_config* cfg, long long* lo_hi, int capacity);
int vu_model_backward_units(const vu_config* cfg, const float* params, const void* shadow,
                            const float* bn_state, float* grads, const float* dy, float* dx, void* ws,
                            size_t ws_bytes, int B, int training, uint64_t seed, const uint32_t* rng_salt,
                            int first_unit, int last_unit, void* stream);

/* ---- per-op entry points (used by the parity tests and by the stand-alone sub-modules) ---- */

/* patch / unpatch / downsampling / upsampling (model.py:8-53): one latent image re-tiled from
 * patch size s_in to s_out (the (B,C,im,im) image is the s = im case).  pos (fp32, output
 * layout, one sample) is added when non-NULL (PatchEncoder, model.py:84-91). */
int vu_retile(int dtype, int in_f32, int out_f32, const void* in, void* out, const float* pos,
              int B, int C, int im, int s_in, int s_out, void* stream);
/* out = retile(in) + add: the backward of a level change where a second gradient arrives in the output tiling - the gradient of
 * downsampling (model.py:40-53 run backwards) meets the one that comes through the skip connection (model.py:407-410, 420-423).  All
 * three tensors in the storage type, `add` and `out` in the s_out tiling, three distinct buffers; the sum is formed in fp32 and
 * rounded once, as a separate add would leave it. */
int vu_retile_add(int dtype, const void* in, const void* add, void* out, int B, int C, int im, int s_in, int s_out, void* stream);

/* Conv2d(C,C,3,padding='same') applied per patch (model.py:137-139,152-154) or on the whole
 * image (model.py:370,428: npatch = B, s = im). */
int vu_conv3x3_fwd(int dtype, int out_f32, const void* in, const float* w, const float* bias,
                   void* out, long long npatch, int C, int s, void* stream);
int vu_conv3x3_bwd(int dtype, int dout_f32, const void* dout, const void* in, const float* w,
                   const void* add, void* din, float* dw, float* dbias, long long npatch, int C,
                   int s, void* stream);

/* The three convolutions of a ReAttention / SkipConnection at once (model.py:137-139 qconv2d / kconv2d / vconv2d applied at
 * :152-154; SkipConnection :246-248 takes q from one tensor and k, v from another: xq != xkv): one read of the input(s),
 * no bias; and the sum of their data gradients: dxq = convT(dq) [+ convT(dk) + convT(dv) when dxkv is NULL] + add_q,
 * dxkv = convT(dk) + convT(dv) + add_kv (add_* may be NULL).  Stencil kernels (csrc/vu_conv.hip); a matrix-core form exists
 * behind VU_CONV_MM=1 (csrc/vu_conv_mm.hip: measured, not faster). */
int vu_conv3x3_qkv_fwd(int dtype, const void* xq, const void* xkv, const float* wq, const float* wk, const float* wv,
                       void* q, void* k, void* v, long long npatch, int C, int s, void* stream);
int vu_conv3x3_qkv_dgrad(int dtype, const void* dq, const void* dk, const void* dv, const float* wq, const float* wk,
                         const float* wv, const void* add_q, const void* add_kv, void* dxq, void* dxkv,
                         long long npatch, int C, int s, void* stream);
/* ... and their weight gradients (autograd of model.py:152-154): dwq += d(q)/d(wq) etc., fp32 (C, C, 3, 3).  `scratch` (optional, at
 * least 1 MiB, device memory): with it the per-workgroup partial sums are added in a fixed order (bit-reproducible; the model executor
 * always lends one), without it the sums end in float atomics. */
int vu_conv3x3_qkv_wgrad(int dtype, const void* dq, const void* dk, const void* dv, const void* xq, const void* xkv, float* dwq,
                         float* dwk, float* dwv, void* scratch, size_t scratch_bytes, long long npatch, int C, int s, void* stream);

/* ReAttention.forward / SkipConnection.forward (model.py:150-164, 244-259) on token maps
 * xq, xkv (B,N,D); prm = pointers into the arenas.  ws from vu_attn_workspace_bytes. */
typedef struct vu_attn_params {
  const float* mix_w; const float* mix_b; const float* bn_w; const float* bn_b;
  const float* wq; const float* wk; const float* wv;
  const void* proj_w;  /* storage dtype */
  const float* proj_b;
  float* run_mean; float* run_var;
  int operands;  /* 0 storage dtype, 1 OCP e4m3 (as vu_config.attn_operands) */
} vu_attn_params;
typedef struct vu_attn_grads {
  float* mix_w; float* mix_b; float* bn_w; float* bn_b; float* wq; float* wk; float* wv;
  float* proj_w; float* proj_b;
} vu_attn_grads;
size_t vu_attn_workspace_bytes(int dtype, int B, int N, int D, int H);
int vu_attn_forward(int dtype, const vu_attn_params* prm, const void* xq, const void* xkv, void* y,
                    void* map_out, void* ws, size_t ws_bytes, int B, int N, int D, int H, int C,
                    float attn_drop, float proj_drop, int training, uint64_t seed, uint64_t stream_id,
                    void* stream);
int vu_attn_backward(int dtype, const vu_attn_params* prm, const vu_attn_grads* grd, const void* xq,
                     const void* xkv, const void* dy, void* dxq, void* dxkv, void* ws, size_t ws_bytes,
                     int B, int N, int D, int H, int C, float attn_drop, float proj_drop, int training,
                     uint64_t seed, uint64_t stream_id, void* stream);

/* residual add + LayerNorm((N,D)) (model.py:193-196, 203-206); x may be NULL. */
size_t vu_layernorm_workspace_floats(int B, long long P);
int vu_add_layernorm_fwd(int dtype, const void* a, const void* x, void* z, const float* w,
                         const float* b, void* y, float* ws, float* stats, int B, long long P,
                         void* stream);
int vu_layernorm_bwd(int dtype, const void* dy, const void* z, const float* w, const float* stats,
                     float* dw, float* db, float* ws, void* dz, int B, long long P, void* stream);

/* FeedForward.forward (model.py:95-110): y = Dropout(Linear(Dropout(GELU(Linear(x))))) on (rows, D) tokens; w1 (hid,D),
 * w2 (D,hid) in the storage dtype, biases fp32.  hpre / hact (rows,hid) keep the GELU pre-activation and the (dropped)
 * activation for the backward.  vu_ff_backward accumulates dw1 (hid,D), db1, dw2 (D,hid), db2 (fp32) and writes dx;
 * scratch >= vu_ff_scratch_bytes.  Dropout sites replay from (seed, stream_id) as in vu_attn_forward. */
size_t vu_ff_scratch_bytes(int dtype, long long rows, int D, int hid);
int vu_ff_forward(int dtype, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                  void* hpre, void* hact, void* y, long long rows, int D, int hid, float linear_drop,
                  int training, uint64_t seed, uint64_t stream_id, void* stream);
int vu_ff_backward(int dtype, const void* x, const void* w1, const void* w2, const void* hpre, const void* hact,
                   const void* dy, void* dx, float* dw1, float* db1, float* dw2, float* db2, void* scratch,
                   long long rows, int D, int hid, float linear_drop, int training, uint64_t seed,
                   uint64_t stream_id, void* stream);

/* strided batched GEMM C = alpha * A * B (+bias) used for every contraction on the path;
 * strides in elements; exactly one of (sAm,sAk) and one of (sBk,sBn) must be 1. */
int vu_gemm(int dtype, int c_float, const void* A, const void* Bm, void* C, int M, int N, int K,
            long long sAm, long long sAk, long long sBk, long long sBn, long long ldc, int Z1, int Z2,
            long long sA1, long long sA2, long long sB1, long long sB2, long long sC1, long long sC2,
            float alpha, const float* bias, int accumulate, void* stream);

/* MSELoss + its gradient (run_denoising.py:80); partials: >= 1024 floats. */
int vu_mse_loss(const float* out, const float* target, float* dout, float* loss, float* partials,
                long long n, float grad_scale, void* stream);
/* AdamW over the flat arena (run_denoising.py:81); hyper = {lr,beta1,beta2,eps,weight_decay}
 * and the int step counter live in device memory. */
int vu_adamw(float* params, const float* grads, float* m, float* v, void* shadow_bf16, long long n,
             const float* hyper, int* step, float grad_scale, void* stream);
int vu_cast_bf16(const float* in, void* out, long long n, void* stream);
/* out[n] += sum over rows of in[row * ld + n], n < ncols (bias / embedding gradients: the sum over the batch). */
int vu_colsum(int dtype, const void* in, float* out, long long rows, int ncols, long long ld, void* stream);
/* In place: every element of x (dtype 0 fp32 / 1 bf16, n % 4 == 0) becomes the nearest OCP e4m3fn value (round to
 * nearest even, 3 mantissa bits, subnormal step 2^-9, saturating at +-448; NaN stays NaN) held in the same container.
 * An e4m3 value is exactly a bf16 value, so the bf16 MFMA on rounded operands forms the same products an fp8 MFMA
 * does; at this model's head dims (d <= 96) the non-scaled fp8 MFMA has the bf16 rate, so nothing is lost.
 * The conversion itself is the hardware's (v_cvt_pk_fp8_f32 / v_cvt_f32_fp8). */
int vu_round_e4m3(int dtype, void* x, long long n, void* stream);

/* Dice loss with its gradient (README.md:91-101: smooth = 1, flattened,
 * 1 - (2 sum(p t) + 1) / (sum p + sum t + 1)).  apply_sigmoid != 0: p = sigmoid(logits) (the
 * segmentation head of BASELINE config 5 - the reference has none) and dlogits is the
 * gradient w.r.t. the logits; dlogits may be NULL (loss only).  partials: >=
 * vu_dice_partials_floats() floats.  Operands 16-byte aligned. */
size_t vu_dice_partials_floats(void);
int vu_dice_loss(const float* logits, const float* target, float* dlogits, float* loss,
                 float* partials, long long n, int apply_sigmoid, float grad_scale, void* stream);
/* Per-image PSNR = 10 log10(R^2 / mean((target-out)^2)) over P elements per image
 * (vit_unet/torch/functions.py:7-19, skimage.metrics.peak_signal_noise_ratio per image);
 * psnr: B floats; partials: >= vu_psnr_partials_floats(B) floats. */
size_t vu_psnr_partials_floats(int B);
int vu_psnr(const float* target, const float* out, float* psnr, float* partials, int B, long long P,
            float data_range, void* stream);
/* Per-image mean SSIM of (B,C,H,W) float images, averaged over channels (README.md:85-89;
 * scikit-image structural_similarity defaults: uniform win x win window (7), K1 = 0.01,
 * K2 = 0.03, sample covariance, (win-1)/2 border cropped).  ssim: B floats; partials: >=
 * vu_ssim_partials_floats(...) floats. */
size_t vu_ssim_partials_floats(int B, int C, int H, int W, int win);
int vu_ssim(const float* target, const float* out, float* ssim, float* partials, int B, int C, int H,
            int W, int win, float data_range, void* stream);

/* Device-side input pipeline of the denoising runs: DenoisingDataset.__getitem__
 * (vit_unet/torch/dataset.py:52-71) + the albumentations transforms of run_denoising.py:52-59 for
 * a whole batch.  noisy / clean: (B,Hs,Ws,channels) uint8 HWC as decoded; x / y: (B,channels,im,im)
 * float32.  Steps: cv2.resize to im x im (skipped when Hs == Ws == im) -> warpAffine with the
 * per-image INVERSE 2x3 matrices inv_affine (B x 6 doubles, device memory; NULL = validation
 * transform, no warp; bilinear for noisy, nearest for clean, constant border 0) -> Normalize
 * ((v - 255 mean) / (255 std)) on noisy only -> /255 on both -> CHW.  A pair may be omitted
 * (noisy == x == NULL or clean == y == NULL).  scratch: >= vu_denoise_prepare_scratch_bytes. */
size_t vu_denoise_prepare_scratch_bytes(int B, int im, int channels);
int vu_denoise_prepare(const uint8_t* noisy, const uint8_t* clean, float* x, float* y,
                       uint8_t* scratch, size_t scratch_bytes, const double* inv_affine, int B,
                       int Hs, int Ws, int channels, int im, float mean, float std, void* stream);

/* Device-side input pipeline for SegmentationDataset items (vit_unet/torch/dataset.py:18-38): the
 * reference reads a DICOM slice (pixel_array, 16-bit) and one NIfTI mask plane on the host and
 * leaves scaling/augmentation to the caller's `augments`.  Here, for a whole batch: image
 * (B,Hs,Ws) int16 and mask (B,Hs,Ws) uint8 labels -> cv2.resize to oh x ow (bilinear, float
 * coefficients, for the slice; nearest for the mask; skipped when sizes match) -> optional
 * warpAffine with per-image INVERSE 2x3 matrices (B x 6 doubles, device; NULL = no warp;
 * bilinear / nearest, border 0) -> x = clip((v - lo) / (hi - lo), 0, 1), y = label * (1 - ls) +
 * ls / 2 (the dataset's `ls` argument); x / y: (B,1,oh,ow) float32.  Either pair may be omitted
 * (is_test).  scratch: >= vu_seg_prepare_scratch_bytes, 2-byte aligned. */
size_t vu_seg_prepare_scratch_bytes(int B, int oh, int ow);
int vu_seg_prepare(const int16_t* image, const uint8_t* mask, float* x, float* y, uint8_t* scratch,
                   size_t scratch_bytes, const double* inv_affine, int B, int Hs, int Ws, int oh,
                   int ow, float lo, float hi, float ls, void* stream);

/* ---- ops of the reference's Keras re-implementation (csrc/vu_tfops.hip).  Not on the benchmarked path.
 * n % 4 == 0 everywhere; dropout masks replay from (seed, stream_id) with element index = linear index (vu_dropout) or
 * row * ld + j (softmax rows), as every other dropout site of this library. ---- */
/* out = a + b: the input residual Y = X + unpatch(...) of /root/reference/vit_unet/tf/model.py:208 */
int vu_add(int dtype, const void* a, const void* b, void* out, long long n, void* stream);
/* out = dropout_p(in) (tf/functions.py:176,179 FeedForward.Drop1 / Drop2); the backward is the same call on the gradient */
int vu_dropout(int dtype, const void* in, void* out, long long n, float p, uint64_t seed, uint64_t stream_id, void* stream);
/* exact (erf) GELU and its backward dx = dy * gelu'(x) (tf.keras.activations.gelu, tf/functions.py:175,178) */
int vu_gelu_fwd(int dtype, const void* x, void* y, long long n, void* stream);
int vu_gelu_bwd(int dtype, const void* x, const void* dy, void* dx, long long n, void* stream);
/* Resampling 'max' (mode 0) / 'avg' (mode 1), tf/functions.py:101-124, pool_size 4: (B,N,P) -> (B,N/4,P),
 * y[b, w + s N/8, :] = pool of x[b, 8 w + 2 s + {0,1,4,5}, :] (+ pos[m, :] if pos != NULL: the layer's position embedding). */
int vu_token_pool4_fwd(int dtype, int mode, const void* x, const float* pos, void* y, int B, int N, int P, void* stream);
int vu_token_pool4_bwd(int dtype, int mode, const void* x, const void* dy, void* dx, int B, int N, int P, void* stream);
/* Keras MultiHeadAttention probabilities (tf/functions.py:288-293): p = softmax(scale * s) over the n keys of each of `rows`
 * rows (leading dimension ld), pd = dropout(p).  Backward: ds = scale * p * (g - sum_k p g), g = mask / keep * dpd. */
int vu_softmax_rows_fwd(int dtype, const void* s, void* p, void* pd, long long rows, int n, int ld, float scale, float p_drop,
                        uint64_t seed, uint64_t stream_id, void* stream);
int vu_softmax_rows_bwd(int dtype, const void* p, const void* dpd, void* ds, long long rows, int n, int ld, float scale,
                        float p_drop, uint64_t seed, uint64_t stream_id, void* stream);
/* vu_add_layernorm_fwd with the epsilon as an argument (Keras LayerNormalization: 1e-3, per token: B = rows, P = features) */
int vu_add_layernorm_fwd_eps(int dtype, const void* a, const void* x, void* z, const float* w, const float* b, void* y,
                             float* ws, float* stats, int B, long long P, float eps, void* stream);

/* Which form the re-attention runs in: PROCESS-GLOBAL test / experiment setting (the two forms draw different dropout
 * masks, so a forward and its backward - and a workspace carved for them - must see the same value: set it between
 * steps, never between a forward and its backward).
 *   flash    -1: model path per level by the fill rule (recompute form when the launch fills the chip), stand-alone
 *                op materialised (default);  0: never the recompute form;  1: the recompute form wherever covered
 *   centered  1: the stand-alone op uses the model path's centred-map form when the map itself is not asked for
 * Initial values: the environment variables VU_ATTN_FLASH (0 / 1) and VU_ATTN_CENTERED, read once. */
int vu_set_attn_form(int flash, int centered);
/* Recompute form only: how many waves of a workgroup share one 16-row tile and split the streamed keys / queries between them
 * (csrc/vu_flash.hip): 0 = the default (by launch size: 1, or 3 where the unsplit sweeps would leave one wave per SIMD), 1 = unsplit,
 * 2 = wave pairs in 4-wave workgroups, 3 = wave pairs in 8-wave workgroups (the Base / Large level-2 shape).  Process-level, for
 * tests and measurements; results differ between the forms by the order of the fp32 sums only. */
int vu_set_flash_key_split(int ks);
/* Recompute form, 8 heads: probability cache (csrc/vu_flash.hip "probability cache").  1: the training forward's moments sweep stores
 * the packed sign-tagged bf16 probabilities of every 16 x 16 tile (B h N^2 2 bytes per attention module, part of the workspace) and
 * the apply / dq / dk / dv sweeps stream them instead of rebuilding logits -> exp2 -> mask -> pack; 0: every sweep recomputes (the
 * round 2 - 4 form); -1: the build's default.  Results are bit-identical either way.  Process-level like vu_set_attn_form: it changes
 * vu_model_workspace_bytes / vu_attn_workspace_bytes, so set it before sizing a workspace, never between a forward and its backward.
 * Initial value: VU_FLASH_PCACHE (0 / 1), read once. */
int vu_set_flash_pcache(int on);
/* Byte budget of ONE model workspace for those caches (sum over the attention modules): modules get a cache in execution order while
 * the sum fits, the others recompute (bit-identical).  Default 96 GiB; initial value VU_FLASH_PCACHE_BUDGET_MB, read once.  Same rule
 * as vu_set_flash_pcache: set it before sizing a workspace, never between a forward and its backward. */
int vu_set_flash_pcache_budget(unsigned long long bytes);
/* Weight-gradient tails (csrc/vu_gemm.h: vu_defred).  1 (default): inside vu_model_backward / _backward_units the fixed-order sums over
 * per-workgroup partial rows that end the q / k / v convolutions' weight gradients and the head-mix gradients of the materialised map
 * backward are queued and run as ONE launch at the end of the call (before anything outside it reads the gradients); 0: every kernel
 * launches its own reduce at once.  Same sums in the same order: bit-identical gradients.  Initial value VU_DEFER_RED, read once. */
int vu_set_deferred_reductions(int on);

/* Data-parallel gradient exchange over RCCL (csrc/vu_dp.cpp).  The reference has no collective
 * (/root/reference/run_denoising.py:79,87: one 'cuda' device); the default exchange of this build is torch.distributed (backend
 * "nccl" = RCCL) and these are the same sums for a host without torch.  One communicator per process (one process per GPU), created
 * on the current HIP device.  RCCL is opened with dlopen at vu_dp_unique_id / vu_dp_init (the copy the process already holds, or
 * librccl.so.1 of the ROCm installation): the library has no link-time dependency on it.
 *   vu_dp_unique_id         rank 0 fills 128 bytes (an ncclUniqueId) and hands them to every rank by any channel
 *   vu_dp_init              collective over all `world` ranks; one communicator per process (vu_dp_finalize before another)
 *   vu_dp_allreduce_bucket  in-place SUM over the ranks of ptr[0 .. count), dtype 0 = fp32 / 1 = bf16, enqueued on `stream`; the
 *                           1 / world average is applied by vu_adamw's grad_scale
 *   vu_dp_world             ranks of the communicator (0: none)
 * PROCESS-GLOBAL state (the communicator); not thread-safe against itself. */
int vu_dp_unique_id(void* out128);
int vu_dp_init(int rank, int world, const void* unique_id_128);
int vu_dp_allreduce_bucket(void* ptr, long long count, int dtype, void* stream);
int vu_dp_world(void);
int vu_dp_finalize(void);

/* In-process launch profiler (bench.py's roofline leg): PROCESS-GLOBAL state, meant for one
 * instrumented stream at a time.  After vu_prof_enable(stream) an event is
 * recorded behind every launch; vu_prof_report() stops, waits, and returns a JSON object
 * {"<kernel tag>": {"count","ms","flops","bytes"}} with ALGORITHMIC flops / bytes per tag. */
int vu_prof_enable(void* stream);
const char* vu_prof_report(void);
/* Holds the profiled stream for `usec` microseconds (0 .. 200000) with a one-wave kernel, so that the launches the host enqueues
 * behind it execute back to back: without it an interval between two events also contains the time the GPU waited for the host
 * (small batches: launches of 5 - 20 us are faster than the eager host).  Its own interval is not reported.  The report's
 * "flops_strict" counts the model's own products only (no recomputation, no padding); "flops" is the launch's work. */
int vu_prof_gate(int usec);

#ifdef __cplusplus
}
#endif
#endif
