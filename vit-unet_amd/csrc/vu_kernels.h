// Host-side launchers for the non-GEMM kernels (all HBM-bound; see DESIGN.md for the byte model).
// dtype: 0 = fp32 storage, 1 = bf16 storage.  Every launcher enqueues on `st` and returns VU_*.
#pragma once
#include "vu_common.h"
#include "vu_gemm.h"

int vu_gemm_launch(int dtype, int c_float, vu_gemm_args g, hipStream_t st);

// a1-a4: token re-tiling of one latent image between patch sizes (image layout = patch size im).
// in_f32 / out_f32: the tensor is fp32 regardless of dtype (model input / output side).
// (rows, H, dh_src) -> (rows, H, dh_dst) bf16, per head: the first min(dh_src, dh_dst) features copied, the rest of dst zero (head dims
// that are multiples of 4).  n <= 4 tensors in one launch.  The recompute attention's zero-padded operands (vu_model.hip, flash_padded).
int vu_k_head_pad(const void* const* src, void* const* dst, int n, long long rows, int H, int dh_src, int dh_dst, hipStream_t st);
int vu_k_retile(int dtype, int in_f32, int out_f32, const void* in, void* out, const float* pos,
                int B, int C, int im, int s_in, int s_out, hipStream_t st, const void* add = nullptr);      // add: storage type, OUTPUT tiling, out = retile(in) + add
// out[r] += sum_b in[b*P + r]   (positional-embedding gradient)
int vu_k_batch_sum(int dtype, const void* in, float* out, int B, long long P, hipStream_t st);

// K4 / K15: 3x3 convolution with zero halo at the patch border; tensors are (npatch, C, s, s).
int vu_k_conv3x3_fwd(int dtype, int out_f32, const void* in, const float* w, const float* bias,
                     void* out, long long npatch, int C, int s, hipStream_t st);
int vu_k_conv3x3_dgrad(int dtype, int dout_f32, const void* dout, const float* w, const void* add,
                       void* din, long long npatch, int C, int s, hipStream_t st);
int vu_k_conv3x3_wgrad(int dtype, int dout_f32, const void* dout, const void* in, float* dw,
                       float* dbias, long long npatch, int C, int s, hipStream_t st);

// fused q/k/v forms (vu_conv.hip): one read of x gives q,k,v; the three data gradients are summed
int vu_k_conv3x3_qkv_fwd(int dtype, const void* xq, const void* xkv, const float* wq, const float* wk, const float* wv,
                         void* q, void* k, void* v, long long npatch, int C, int s, hipStream_t st);
int vu_k_conv3x3_qkv_dgrad(int dtype, const void* dq, const void* dk, const void* dv, const float* wq, const float* wk,
                           const float* wv, const void* add_q, const void* add_kv, void* dxq, void* dxkv,
                           long long npatch, int C, int s, hipStream_t st);
// the same two in the Toeplitz form on the matrix cores (vu_conv_tz.hip, round 6): bf16, C = 3, patch size 16 / 8 - the default route
bool vu_conv_tz_ok(int dtype, int C, int s, long long npatch);
int vu_k_conv_tz_qkv_fwd(const void* xq, const void* xkv, const float* wq, const float* wk, const float* wv, void* q, void* k, void* v,
                         long long npatch, int s, hipStream_t st);
int vu_k_conv_tz_qkv_dgrad(const void* dq, const void* dk, const void* dv, const float* wq, const float* wk, const float* wv,
                           const void* add_q, const void* add_kv, void* dxq, void* dxkv, long long npatch, int s, hipStream_t st);
size_t vu_conv_tz_wgrad_scratch_floats();       // per-workgroup partial sums of the weight-gradient form below
int vu_k_conv_tz_qkv_wgrad(const void* dq, const void* dk, const void* dv, const void* xq, const void* xkv, float* dwq, float* dwk, float* dwv,
                           float* part, long long npatch, int s, hipStream_t st, int defer = 0);      // defer: `part` came from vu_deferred_take, queue the reduce
// the same two on the matrix cores (vu_conv_mm.hip): bf16, C in {1, 3}, patch size 8 / 16 / 32 - taken by the two launchers above
bool vu_conv_mm_ok(int dtype, int C, int s, int backward);
int vu_k_conv_mm_qkv_fwd(const void* xq, const void* xkv, const float* wq, const float* wk, const float* wv, void* q, void* k, void* v,
                         long long npatch, int C, int s, hipStream_t st);
int vu_k_conv_mm_qkv_dgrad(const void* dq, const void* dk, const void* dv, const float* wq, const float* wk, const float* wv,
                           const void* add_q, const void* add_kv, void* dxq, void* dxkv, long long npatch, int C, int s, hipStream_t st);
int vu_k_conv3x3_qkv_wgrad(int dtype, const void* dq, const void* dk, const void* dv, const void* xq, const void* xkv,
                           float* dwq, float* dwk, float* dwv, long long npatch, int C, int s, hipStream_t st);

// K7+K8: row softmax + dropout on (rows, ld) logits, N valid columns; writes sign-tagged
// probabilities (negative = dropped) in place.
int vu_k_softmax_dropout(int dtype, void* S, long long rows, int N, int ld, vu_rng rng,
                         hipStream_t st);

// fused K6+K7+K8 (vu_attn.hip): logits by MFMA, row softmax, dropout, sign-tagged store; the
// logits never reach HBM.  Returns 1 when the shape is not covered (caller falls back).
int vu_k_attn_scores(int dtype, const void* q, const void* k, void* Ps, int B, int N, int D, int H, int ld,
                     float scale, vu_rng rng, hipStream_t st);

// map x head-slice products for long rows / small head dims (streaming MFMA kernels); 1 = shape not covered
int vu_k_attn_map_prod(int dtype, int cols, const void* M, const void* X, void* out, const float* sc, const float* kappa,
                       int B, int N, int D, int H, int ld, hipStream_t st);
// short rows, 8 heads (vu_attn_fused.hip, round 6): scores + softmax + dropout + head mix + shifted moments in ONE launch; Ps and the
// centred map Ac as vu_k_attn_scores + vu_k_mix_stats_mm(Ac) leave them, *nblocks rows of partials for vu_k_bn_finalize
bool vu_attn_f1_ok(int dtype, int B, int N, int D, int H, int ld);
int vu_k_attn_f1(const void* q, const void* k, void* Ps, void* Ac, const float* W, float* partials, int* nblocks, int B, int N, int D, int ld,
                 float scale, vu_rng rng, hipStream_t st);
int vu_k_attn_outer(int dtype, const void* a, const void* bmat, void* out, int B, int N, int D, int H, int ld,
                    float scale, hipStream_t st);

// K9+K10: head mixing + BatchNorm on sign-tagged maps (B,H,N,ld).
// stats buffer layout (floats): Wf[H*H] cf[H] mean[H] rstd[H] m1[H] m2[H]
// backward tables appended: X[H*H] = W*rstd_g, Xc[H] = (c-mean)*rstd, Gs[H] = gamma*rstd
// then sc[H], kappa[H] of the centred-map form, then the tables of the non-materialising form (vu_flash.hip):
// FWk[H*H] = gamma rstd W / keep, XK[H*H] = rstd W / keep
#define VU_BN_STATS_FLOATS(H) (4 * (H) * (H) + 10 * (H))
#define VU_BN_STATS_SC(H) (2 * (H) * (H) + 8 * (H))
#define VU_BN_STATS_FWK(H) (2 * (H) * (H) + 10 * (H))
#define VU_BN_STATS_XK(H) (3 * (H) * (H) + 10 * (H))
int vu_k_mix_stats(int dtype, const void* Ps, const float* W, const float* c, float* partials,
                   int nblocks, int B, int H, int N, int ld, float inv_keep, hipStream_t st);
int vu_k_mix_stats_mm(int dtype, const void* Ps, const float* W, float* partials, void* Ac, int nblocks, int B, int H, int N,
                      int ld, float inv_keep, hipStream_t st);   // MFMA form (bf16, H = 8, 256 < ld <= 1024); 1 = not covered
int vu_k_bn_finalize(const float* partials, int nblocks, const float* W, const float* c,
                     const float* gamma, const float* beta, float* run_mean, float* run_var,
                     float* stats, int H, int N, double count, int training, float momentum,
                     float eps, hipStream_t st);
int vu_k_mix_apply(int dtype, const void* Ps, void* Ahat, const float* stats, int B, int H, int N,
                   int ld, float inv_keep, hipStream_t st);
int vu_k_bn_bwd_stats(int dtype, const void* Ps, const void* dAhat, const float* W, const float* c,
                      const float* stats, float* partials, int nblocks, int B, int H, int N, int ld,
                      float inv_keep, hipStream_t st);
int vu_k_bn_bwd_finalize(const float* partials, int nblocks, float* stats, float* dgamma,
                         float* dbeta, int H, double count, int training, hipStream_t st);
// the same statistics from dO, O and v only (no pass over the maps); partials >= B*2*H floats
int vu_k_bn_bwd_small(int dtype, const void* dO, const void* O, const void* v, const float* gamma, const float* beta,
                      const float* W, const float* c, float* stats, float* dgamma, float* dbeta, float* partials, int B, int N, int D, int H,
                      int training, hipStream_t st);
// dS written over dAhat.  dW (H*H) and dc (H) are accumulated with atomics.
int vu_k_map_bwd(int dtype, const void* Ps, void* dAhat_dS, const float* W, const float* c,
                 const float* gamma, const float* stats, float* dW, float* dc, int B, int H, int N,
                 int ld, float inv_keep, float scale, hipStream_t st);
int vu_k_map_bwd_2sweep(int dtype, const void* Ps, void* dAhat_dS, const float* W, const float* c,
                        const float* gamma, const float* stats, float* dW, float* dc, int B, int H, int N,
                        int ld, float inv_keep, float scale, hipStream_t st);

// K13: residual add + LayerNorm over all P elements of a sample.
#ifndef VU_LN_ITS
#define VU_LN_ITS 2          /* 16-byte (8 x bf16) accesses per thread and chunk; the 4-element kernels make twice as many */
#endif
#define VU_LN_CHUNK (2048 * VU_LN_ITS)
#define VU_LN_BCHUNK 256
static inline int vu_ln_nchunks(long long P) { return (int)((P + VU_LN_CHUNK - 1) / VU_LN_CHUNK); }
static inline int vu_ln_nbchunks(long long P) { return (int)((P + VU_LN_BCHUNK - 1) / VU_LN_BCHUNK); }
// partials: B*nchunks*3 floats ; stats: B*2 floats (mean, rstd)
int vu_k_add_ln_fwd(int dtype, const void* a, const void* x, void* z, const float* w,
                    const float* bias, void* y, float* partials, float* stats, int B, long long P,
                    float eps, hipStream_t st);
// partials2: B*nbchunks*2 floats.  dz_drop (optional): dropout(dz) with rng (proj dropout bwd).
int vu_k_ln_bwd(int dtype, const void* dy, const void* z, const float* w, const float* stats,
                float* dw, float* db, float* partials2, void* dz, void* dz_drop, vu_rng rng,
                int B, long long P, hipStream_t st);

// column sums: out[n] += sum_m in[m*ld + n]
int vu_k_colsum(int dtype, const void* in, float* out, long long rows, int ncols, long long ld,
                hipStream_t st);
int vu_k_add(int dtype, const void* a, const void* b, void* out, long long n, hipStream_t st);
int vu_k_dropout(int dtype, const void* in, void* out, long long n, vu_rng rng, hipStream_t st);

// a13: loss / optimizer
int vu_k_mse(const float* out, const float* target, float* dout, float* loss, float* partials,
             long long n, float gscale, hipStream_t st);
int vu_k_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n,
               const float* hyper, int* step, float gscale, hipStream_t st);
int vu_k_cast_bf16(const float* in, void* out, long long n, hipStream_t st);
int vu_k_round_e4m3(int dtype, void* x0, void* x1, void* x2, long long n, hipStream_t st);   // up to three arrays of n elements
int vu_k_fill(float* p, float v, long long n, hipStream_t st);
