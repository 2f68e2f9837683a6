// Entry points of the fused scores kernels (vu_attn_scores.h); the four {type} x {form} instantiation sets live in
// vu_attn_sc_*.hip so that they build in parallel.
#include <stdlib.h>
#include "vu_kernels.h"

#define VU_SC_DECL(name) int name(const void*, const void*, void*, int, int, int, int, int, float, vu_rng, hipStream_t)
VU_SC_DECL(vu_scores_f32_softmax); VU_SC_DECL(vu_scores_f32_plain); VU_SC_DECL(vu_scores_bf16_softmax); VU_SC_DECL(vu_scores_bf16_plain);

// returns VU_OK, a negative error, or 1 when the shape is not covered by the fused kernel
int vu_k_attn_scores(int dtype, const void* q, const void* k, void* Ps, int B, int N, int D, int H, int ld, float scale,
                     vu_rng rng, hipStream_t st) {
  const int d = D / H;
  const int dp = (d + 31) / 32 * 32;
  const size_t es = dtype == 0 ? 4 : 2;
  const size_t lds = (size_t)((N + 15) & ~15) * (dp + (es == 2 ? 8 : 4)) * es;      // (the kernel zero-pads K to whole key tiles)
  if (lds > 150 * 1024 && !(N > 784 && dp == 32)) return 1;      // (long rows stream K in chunks)
  if ((double)B * H * N * (double)ld >= 4294967295.0) return 1;   // 32-bit mask index in the fused kernel
  if (dtype == 0) return vu_scores_f32_softmax(q, k, Ps, B, N, D, H, ld, scale, rng, st);
  return vu_scores_bf16_softmax(q, k, Ps, B, N, D, H, ld, scale, rng, st);
}

// out[b,g,i,j] = scale * sum_t a[b,i,g*d+t] * bmat[b,j,g*d+t]   (dAhat = dO v^T); 1 = shape not covered
int vu_k_attn_outer(int dtype, const void* a, const void* bmat, void* out, int B, int N, int D, int H, int ld, float scale,
                    hipStream_t st) {
  const int d = D / H;
  const int dp = (d + 31) / 32 * 32;
  const size_t es = dtype == 0 ? 4 : 2;
  const size_t lds = (size_t)((N + 15) & ~15) * (dp + (es == 2 ? 8 : 4)) * es;      // (the kernel zero-pads K to whole key tiles)
  if (lds > 150 * 1024 && !(N > 784 && dp == 32)) return 1;
  vu_rng none = vu_make_rng(0, 0, 0.f);
  if (dtype == 0) return vu_scores_f32_plain(a, bmat, out, B, N, D, H, ld, scale, none, st);
  return vu_scores_bf16_plain(a, bmat, out, B, N, D, H, ld, scale, none, st);
}

