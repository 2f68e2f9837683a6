// Non-materialising re-attention (vu_flash.hip): host-side launchers.
#pragma once
#include "vu_common.h"

struct vu_flash_args {
  int B, N, D, H;
  float scale;                 // d^-0.5 (model.py:131)
  int training;                // 1: batch statistics + dropout (rng); 0: running statistics
  vu_rng rng;                  // attention-map dropout stream (thr == 0: no dropout)
  const void *q, *k, *v;       // (B,N,D) bf16
  void* O;                     // (B,N,D) bf16: A^ v, heads re-concatenated (model.py:161)
  float* lse2;                 // (B,H,N): log2-domain log-sum-exp of the scaled logits, kept for the backward
  float* rinv;                 // (B,H,N): 1 / (row sum of the probabilities as every sweep recomputes them from lse2), kept for the backward
  float* pk;                   // (B,N,D) fp32: sum_k bf16(P) k per head (the training forward writes it, the backward's fused delta + dq sweep reads it; null: separate sweeps)
  void* pcache;                // 8-head form, optional: B (N/16)^2 tiles of 4 KB - the sign-tagged bf16 probabilities of every (query tile, key tile) as
                               // the moments sweep packed them (vu_flash_pcache_bytes); the later sweeps stream it instead of rebuilding logits -> exp2 -> mask -> pack
  float* rinv_b;               // (B,H,N) 4-head form only: 1 / sum_k bf16(P) of the moments sweep (the fused dq's delta~; rinv keeps the fp32 row sum for the dk sweep)
  float* partials;             // >= vu_flash_partials_floats()
  float* stats;                // VU_BN_STATS_FLOATS(H): folded tables (vu_kernels.h)
  const float *mix_w, *mix_b, *bn_w, *bn_b;
  float *run_mean, *run_var;
  // backward only
  const void* dO;              // (B,N,D) bf16
  void *dq, *dk, *dv;          // (B,N,D) bf16
  float* delta;                // (B,H,N)
  float *d_mix_w, *d_mix_b;    // accumulated
};

// attention-map dropout of this form: 8 bits per element, drop probability round(256 p) / 256 (vu_flash.hip, "quad" scheme)
vu_rng vu_flash_quad_rng(vu_rng r);
// true when the backward of this shape, launched eagerly (outside a stream capture), overlaps its dv sweep with the tails of the
// dq / dk sweeps on a low-priority stream (vu_flash.hip "Tail overlap")
bool vu_flash_tail_overlap(int B, int N, int H);
// 1 or 2: waves of a workgroup that share one own tile and split the streamed axis (small launches; vu_flash.hip)
int vu_flash_key_split(int B, int N);
bool vu_flash_ok(int dtype, int B, int N, int D, int H);
// the recompute form only pays when its grid (B x ceil(N/64) work groups of 4 waves) puts a work group on most CUs; below that
// the materialising kernels (which parallelise over heads too) are faster.  Measured on Base (N = 784: 13 groups per sample)
// after the round-2 ISA pass: 8 images -9 %, 16 +3 %, 24 +6 %, 32 +4 %, 64 +12 %; Large at 16: +3 %  ->  threshold 192 groups
bool vu_flash_pays(int B, int N);
size_t vu_flash_partials_floats(int B, int N, int H);
// bytes of the probability cache of one module (0: this shape / this process does not use one).  Process-level switch, read once:
// VU_FLASH_PCACHE=0 / 1, vu_set_flash_pcache()
size_t vu_flash_pcache_bytes(int B, int N, int D, int H);
// bytes one model workspace may spend on those caches in total (vu_set_flash_pcache_budget; VU_FLASH_PCACHE_BUDGET_MB)
size_t vu_flash_pcache_budget();
int vu_k_flash_forward(const vu_flash_args& a, hipStream_t st);
int vu_k_flash_backward(const vu_flash_args& a, hipStream_t st);
