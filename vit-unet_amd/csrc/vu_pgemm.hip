// Persistent small-weight linear for bf16 storage:   C[M][N] = epilogue(A[M][K] . W),   N, K <= 192, M = B*N tokens.
// The level-2 linears of the model (192 -> 192 projection and its data gradient, 192 <-> 32 feed-forward layers,
// model.py:95-110,161-163) have M = 50 176 rows and a weight of 12 - 72 KB.  As tiles of the general GEMM every
// workgroup lived for three dependent k-steps (load - wait - multiply) plus an epilogue and re-read the weight from L2:
// 30 us per launch for 58 MB (1.9 TB/s).  Here one workgroup per CU keeps the WHOLE weight in LDS for its lifetime and
// walks over 128-row slabs of A: the A fragments of the next slab are loaded (straight into the MFMA operand layout:
// A is K-contiguous, a lane's 8 k-values are one 16-byte load) while the current slab is multiplied and written, so
// the loads never wait for a workgroup to start.  The epilogue is vu_gemm's (bias, GELU with saved pre-activation,
// GELU', dropout with the same element index, residual addend) on 16-byte row-contiguous accesses through a
// wave-private fp32 LDS tile.
#include <stdio.h>
#include <stdlib.h>
#include "vu_gemm.h"

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_q;

// TBF: the weight operand is row-contiguous in n (B(k,n) = W[k*ldb + n], data gradients) - LDS image [k][n], fragments by
// the transposing read; otherwise K-contiguous (B(k,n) = W[n*ldb + k], forward) - image [n][k], fragments by ds_read_b128.
template <bool TBF, int NT16, int KT32, int TM>
__global__ __launch_bounds__(256) void vu_pgemm_kernel(const vu_gemm_args g) {
  constexpr int N = 16 * NT16, K = 32 * KT32, SLAB = 64 * TM;
  constexpr int LDB = TBF ? N + 16 : K + 8;
  constexpr int B_ELEMS = TBF ? K * LDB : N * LDB;
  constexpr int LDC = N + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];      // B_ELEMS * 2 + 4 * 16 * LDC * 4 bytes
  bf16_t* Bs = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  float* Ct = reinterpret_cast<float*>(smem_raw + B_ELEMS * 2) + wave * 16 * LDC;
  const bf16_t* A = (const bf16_t*)g.A;
  const bf16_t* W = (const bf16_t*)g.B;
  const long long lda = g.sAm, ldb = TBF ? g.sBk : g.sBn;
  // ---- the weight, once ----
  {
    constexpr int ROWS = TBF ? K : N, COLS = TBF ? N : K;      // image rows x contiguous columns
    for (int v = tid; v < ROWS * (COLS / 8); v += 256) {
      const int r = v / (COLS / 8), c = (v - r * (COLS / 8)) * 8;
      *reinterpret_cast<uint4*>(&Bs[r * LDB + c]) = *reinterpret_cast<const uint4*>(W + (long long)r * ldb + c);
    }
  }
  const vu_rng rng = g.dropout ? vu_rng_resolve(g.rng) : g.rng;
  bf16_t* Cb = (bf16_t*)g.C;
  bf16_t* auxb = (bf16_t*)g.aux;
  const bf16_t* addb = (const bf16_t*)g.addend;
  const int nslab = (g.M + SLAB - 1) / SLAB;
  uint4 cur[TM][KT32], nxt[TM][KT32];
  auto fetch = [&](int slab, uint4 (&f)[TM][KT32]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int m = slab * SLAB + wave * 16 * TM + i * 16 + l15;
      if (m >= g.M) m = g.M - 1;                              // clamped (rows past the end are never stored)
      const bf16_t* ap = A + (long long)m * lda + 8 * lg;
#pragma unroll
      for (int kk = 0; kk < KT32; ++kk) f[i][kk] = *reinterpret_cast<const uint4*>(ap + 32 * kk);
    }
  };
  int slab = blockIdx.x;
  if (slab < nslab) fetch(slab, cur);
  __syncthreads();                                            // the weight image is complete
  for (; slab < nslab; slab += gridDim.x) {
    const int nslab_next = slab + gridDim.x;
    if (nslab_next < nslab) fetch(nslab_next, nxt);           // in flight during the products and the epilogue
    f32x4 acc[TM][NT16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NT16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KT32; ++kk) {
#pragma unroll
      for (int j = 0; j < NT16; ++j) {
        bf16x8 bf;
        if constexpr (TBF) {
          const int q = l15 >> 2, p = l15 & 3;
          const bf16_t* b0 = &Bs[(32 * kk + 8 * lg + q) * LDB + 16 * j + 4 * p];
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_q)b0);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_q)(b0 + 4 * LDB));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bf = __builtin_bit_cast(bf16x8, x);
        } else {
          bf = *reinterpret_cast<const bf16x8*>(&Bs[(16 * j + l15) * LDB + 32 * kk + 8 * lg]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur[i][kk]), bf, acc[i][j], 0, 0, 0);
      }
    }
    // ---- epilogue: one 16-row tile at a time through the wave's fp32 tile (wave-private: LDS operations of a wave are
    // ordered, no barrier) ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = slab * SLAB + wave * 16 * TM + i * 16;
#pragma unroll
      for (int j = 0; j < NT16; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(4 * lg + r) * LDC + 16 * j + l15] = acc[i][j][r] * g.alpha;
      constexpr int CH = N / 8;                               // 16-byte chunks per row
#pragma unroll
      for (int it = 0; it < (16 * CH + 63) / 64; ++it) {
        const int c = lane + 64 * it;
        const int rl = c / CH, c8 = (c - rl * CH) * 8;
        const int orow = row0 + rl;
        if (c < 16 * CH && orow < g.M) {
          float v[8];
          const float4 a0 = *reinterpret_cast<const float4*>(&Ct[rl * LDC + c8]), a1 = *reinterpret_cast<const float4*>(&Ct[rl * LDC + c8 + 4]);
          v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
          if (g.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(g.bias + c8), b1 = *reinterpret_cast<const float4*>(g.bias + c8 + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          }
          const long long o = (long long)orow * g.ldc + c8;
          union U8 { uint4 u; bf16_t h[8]; };
          if (g.act == VU_ACT_GELU) {
            U8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) { t.h[e] = (bf16_t)v[e]; v[e] = vu_gelu(v[e]); }
            *reinterpret_cast<uint4*>(auxb + o) = t.u;
          } else if (g.act == VU_ACT_DGELU) {
            U8 t; t.u = *reinterpret_cast<const uint4*>(auxb + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= vu_gelu_grad((float)t.h[e]);
          }
          if (g.dropout) {
            const uint64_t idx = (uint64_t)orow * (uint64_t)N + c8;      // z = 0: element index of vu_gemm's epilogue
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vu_keep(rng, idx + e) ? v[e] * rng.inv_keep : 0.f;
          }
          if (addb) {
            U8 t; t.u = *reinterpret_cast<const uint4*>(addb + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)t.h[e];
          }
          U8 w;
#pragma unroll
          for (int e = 0; e < 8; ++e) w.h[e] = (bf16_t)v[e];
          *reinterpret_cast<uint4*>(Cb + o) = w.u;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int kk = 0; kk < KT32; ++kk) cur[i][kk] = nxt[i][kk];
  }
}

template <bool TBF, int NT16, int KT32>
int launch_p(const vu_gemm_args& g, hipStream_t st) {
  static const int tm_env = getenv("VU_PGEMM_TM") ? atoi(getenv("VU_PGEMM_TM")) : 2;      // measurement switch (64-row slabs measured slower)
  const int TMr = tm_env == 1 ? 1 : 2;
  const int nslab = (g.M + 64 * TMr - 1) / (64 * TMr);
  constexpr size_t lds_ = (size_t)(TBF ? (32 * KT32) * (16 * NT16 + 16) : (16 * NT16) * (32 * KT32 + 8)) * 2 + (size_t)4 * 16 * (16 * NT16 + 4) * 4;
  int grid = lds_ <= 80 * 1024 ? 512 : 256;                     // persistent workgroups: as many as fit the CUs' LDS (two per CU when the weight is small)
  if (grid > nslab) grid = nslab;
  constexpr int N = 16 * NT16, K = 32 * KT32;
  constexpr size_t lds = (size_t)(TBF ? K * (N + 16) : N * (K + 8)) * 2 + (size_t)4 * 16 * (N + 4) * 4;
  auto kern = TMr == 2 ? vu_pgemm_kernel<TBF, NT16, KT32, 2> : vu_pgemm_kernel<TBF, NT16, KT32, 1>;
  static bool reserved[3] = {false, false, false};               // (per instantiation; the attribute is sticky)
  if (!reserved[TMr] && lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vu_set_error("vu_pgemm: cannot reserve %zu bytes of LDS", lds);
      return VU_ELAUNCH;
    }
    reserved[TMr] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, g);
  if (vu_prof_on()) {
    char tag[64];
    snprintf(tag, sizeof(tag), "vu_pgemm_kernel<%s,N%d,K%d>", TBF ? "T" : "N", 16 * NT16, 32 * KT32);
    vu_prof_note(tag, 2.0 * g.M * (double)g.N * g.K,
                 2.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N * (1 + (g.addend ? 1 : 0) + (g.aux ? 1 : 0))));
  }
  return vu_check_launch("vu_pgemm");
}

}  // namespace

// 1 = launched, 0 = shape not covered (the caller falls through to the tiled GEMM), < 0 = error
int vu_pgemm_try(const vu_gemm_args& g, hipStream_t st) {
  static const bool off = [] { const char* e = getenv("VU_PGEMM"); return e && e[0] == '0'; }();       // A/B switch
  if (off) return 0;
  if (g.Z1 * g.Z2 != 1 || g.accumulate || g.colsum || g.M < 1024) return 0;
  if (g.sAk != 1 || g.sAm % 8 || g.ldc % 8 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15)) return 0;
  if ((g.aux && ((uintptr_t)g.aux & 15)) || (g.addend && ((uintptr_t)g.addend & 15)) || (g.bias && ((uintptr_t)g.bias & 15))) return 0;
  if (g.act != VU_ACT_NONE && !g.aux) return 0;
  const bool nform = g.sBk == 1 && g.sBn % 8 == 0, tform = g.sBn == 1 && g.sBk % 8 == 0;
  if (!nform && !tform) return 0;
  const bool tb = !nform;
#define VU_P(NN, KK) \
  if (g.N == NN && g.K == KK) { const int rc = tb ? launch_p<true, NN / 16, KK / 32>(g, st) : launch_p<false, NN / 16, KK / 32>(g, st); return rc ? rc : 1; }
  // 192 x 192 measured level with the tiled kernel (20 - 22 us either way: the whole product is 150 KB per CU, bound by the
  // load -> multiply -> store latency chain rather than by a rate) and stays there unless VU_PGEMM=2
  static const bool all = [] { const char* e = getenv("VU_PGEMM"); return e && e[0] == '2'; }();
  if (all) { VU_P(192, 192) }
  VU_P(32, 192) VU_P(192, 32)
#undef VU_P
  return 0;
}
