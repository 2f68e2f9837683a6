// GEMM instantiations + host launcher.
#include <stdio.h>
#include <stdlib.h>
#include "vu_gemm.h"

template <typename T, typename TC, bool TA, bool TB, int BM, int BN, int BK, int PD = 1>
static int launch_bk(const vu_gemm_args& g, hipStream_t st) {
  vu_gemm_args ga = g;
  const long long blocks = (long long)vu_cdiv(g.M, BM) * vu_cdiv(g.N, BN) * g.Z1 * g.Z2;
  ga.ksplit = 1;
  if (sizeof(TC) == 4 && g.accumulate && !g.act && !g.dropout && !g.addend && blocks < 512) {
    // small output, long K (weight gradients over B*N rows): split K so the chip is filled
    // every split adds its whole tile with float atomics (chip-wide ~1.3 TB/s of added bytes): aim for ~2 blocks
    // per CU and at least 8 k-steps per block rather than for the most blocks
    int want = (int)((512 + blocks - 1) / blocks);
    const int maxs = vu_cdiv(g.K, 8 * BK);
    ga.ksplit = want < maxs ? want : maxs;
    if (ga.ksplit < 1) ga.ksplit = 1;
  }
  dim3 grid((unsigned)(vu_cdiv(g.M, BM) * vu_cdiv(g.N, BN)), (unsigned)(g.Z1 * g.Z2), (unsigned)ga.ksplit);
  hipLaunchKernelGGL((vu_gemm_kernel<T, TC, TA, TB, BM, BN, BK, PD>), grid, dim3(256), 0, st, ga);
  if (vu_prof_on()) {
    char tag[96];
    static const bool shapes = getenv("VU_PROF_SHAPES") != nullptr;
    if (shapes)
      snprintf(tag, sizeof(tag), "gemm<%s,%s,%c%c,%dx%d> M%d N%d K%d Z%d s%d%s", sizeof(T) == 2 ? "bf16" : "f32",
               sizeof(TC) == 2 ? "bf16" : "f32", TA ? 'T' : 'N', TB ? 'T' : 'N', BM, BN, g.M, g.N, g.K, g.Z1 * g.Z2, ga.ksplit,
               g.swap ? " swapped" : "");
    else
      snprintf(tag, sizeof(tag), "vu_gemm_kernel<%s,%s,%c%c,%dx%d>", sizeof(T) == 2 ? "bf16" : "f32",
               sizeof(TC) == 2 ? "bf16" : "f32", TA ? 'T' : 'N', TB ? 'T' : 'N', BM, BN);
    const double Z = (double)g.Z1 * g.Z2;
    vu_prof_note(tag, 2.0 * g.M * (double)g.N * g.K * Z,
                 Z * (((double)g.M * g.K + (double)g.K * g.N) * sizeof(T) + (double)g.M * g.N * sizeof(TC)));
  }
  return vu_check_launch("vu_gemm");
}

template <typename T, typename TC, bool TA, bool TB, int BM, int BN>
static int launch_one(const vu_gemm_args& g, hipStream_t st) {
  // k-tile depth: bf16 tiles run fewer, longer k-steps (one barrier pair per BK); the narrow
  // 128x32 tile (map x small-matrix products) takes BK = 128.  fp32 keeps 32 (LDS size).
  // measured (tools/gemm_bench.py, 3072-class GEMMs): BK 64 is +10..12 % over BK 32
  if constexpr (sizeof(T) == 4) return launch_bk<T, TC, TA, TB, BM, BN, 32>(g, st);
  else if constexpr (BM == 64 && BN == 64) {
    // long K on a small tile: deep register ring (vu_gemm.h); short K keeps the plain prefetch (fewer registers, no gain)
    static const bool ring_off = [] { const char* e = getenv("VU_GEMM_RING"); return e && e[0] == '0'; }();      // A/B switch
    if (g.K >= 512 && !ring_off) return launch_bk<T, TC, TA, TB, BM, BN, 64, 4>(g, st);
    return launch_bk<T, TC, TA, TB, BM, BN, 64>(g, st);
  } else return launch_bk<T, TC, TA, TB, BM, BN, 64>(g, st);
}

template <typename T, typename TC, bool TA, bool TB>
static int launch_tiles(const vu_gemm_args& g, hipStream_t st) {
  static const int force = getenv("VU_GEMM_TILE") ? atoi(getenv("VU_GEMM_TILE")) : 0;     // measurement switch
  if (force == 12864) return launch_one<T, TC, TA, TB, 128, 64>(g, st);
  if (g.N <= 32) return launch_one<T, TC, TA, TB, 128, 32>(g, st);
  if (g.M <= 64) {
    // (the swapped 768 -> 64 layers: 1 x 196 tiles of 64 x 64 over K = 768: halve the row tile so that every CU streams)
    if constexpr (sizeof(T) == 2 && sizeof(TC) == 2) {
      static const bool half_off = [] { const char* e = getenv("VU_GEMM_HALFROW"); return e && e[0] == '0'; }();     // A/B switch
      if (!half_off && g.M > 32 && g.K >= 512 && (long long)vu_cdiv(g.N, 64) * g.Z1 * g.Z2 < 256) {
        // k-steps of 256, one tile in flight (see the 32 x 32 route below): 18.2 -> 15.8 us at 16 images, 20.8 -> 18.9 at 64; VU_GEMM_3264_BK = 64 / 128: the others
        static const int bk3264 = [] { const char* e = getenv("VU_GEMM_3264_BK"); return e ? atoi(e) : 256; }();
        if (bk3264 == 256 && g.K % 256 == 0) return launch_bk<T, TC, TA, TB, 32, 64, 256, 1>(g, st);
        if (bk3264 == 128 && g.K % 128 == 0) return launch_bk<T, TC, TA, TB, 32, 64, 128, 2>(g, st);
        return launch_bk<T, TC, TA, TB, 32, 64, 64, 4>(g, st);
      }
    }
    return launch_one<T, TC, TA, TB, 64, 64>(g, st);
  }
  {
    // many small batched products with one k-step (the per-(sample, head) map x head-slice products of level 0,
    // 49 x 49 x 384): half-height tiles, twice the workgroups (26 -> 23 us, 24 -> 21 us)
    static const bool smallk_off = [] { const char* e = getenv("VU_GEMM_SMALLK"); return e && e[0] == '0'; }();     // A/B switch
    if (!smallk_off && g.N <= 64 && g.K <= 64 && (long long)g.Z1 * g.Z2 >= 64) return launch_one<T, TC, TA, TB, 64, 64>(g, st);
  }
  if (g.N <= 64) return launch_one<T, TC, TA, TB, 128, 64>(g, st);     // (smaller row tiles for the 98-tile 768 -> 64 layers measured no gain)
  // fewer big tiles than CUs and a long K (the 3072 -> 128 feed-forward layers of level 0: 25 tiles of 128x128; the 3072 x 3072
  // projections at 16 images per GPU: 168 tiles): quarter tiles
  // put 4x the blocks on the chip (measured 77 -> see tools/gemm_bench.py); split-K is not available for bf16 outputs
  static const int quarter_below = getenv("VU_GEMM_QUARTER_BELOW") ? atoi(getenv("VU_GEMM_QUARTER_BELOW")) : 200;   // measurement switch (64 -> 200: +3 % on the 16-image-per-GPU steps of Base / Large, whose 3072-wide linears have 168 tiles)
  if (sizeof(TC) == 2 && g.K >= 1024 && (long long)vu_cdiv(g.M, 128) * vu_cdiv(g.N, 128) * g.Z1 * g.Z2 < quarter_below) {
    // still fewer 64 x 64 tiles than half the CUs (the 3072 -> 128 layers of level 0: 98): one CU sustains ~20 GB/s of
    // loads, so halve the tile again
    if constexpr (sizeof(T) == 2) {
      static const bool eighth_off = [] { const char* e = getenv("VU_GEMM_EIGHTH"); return e && e[0] == '0'; }();     // A/B switch
      if (!eighth_off && (long long)vu_cdiv(g.M, 64) * vu_cdiv(g.N, 64) * g.Z1 * g.Z2 < 128) {
        static const bool t3232_off = [] { const char* e = getenv("VU_GEMM_3232"); return e && e[0] == '0'; }();      // A/B switch
        // (98 tiles of 64 x 64: 32 x 32 tiles put 392 workgroups on the chip: 31 -> 26 (32 x 64) -> 23 us)
        if (!t3232_off && (long long)vu_cdiv(g.M, 32) * vu_cdiv(g.N, 64) * g.Z1 * g.Z2 < 256) {
          // 48 k-steps of 64 are 48 barrier pairs on a 4 KB + 4 KB tile: k-steps of 256 with one tile in flight (the same 32 KB per
          // workgroup as the ring of four 64-steps, a quarter of the barriers): 33.1 -> 26.6 us at 16 images, 30.6 -> 26.0 at 64
          // (tools/step_tags.py; 128 x 2: 30.4 / 27.4, 256 x 2: 28.3 / 28.1).  VU_GEMM_3232_BK = 64 / 128 / 257 select the others.
          static const int bk3232 = [] { const char* e = getenv("VU_GEMM_3232_BK"); return e ? atoi(e) : 256; }();
          if (bk3232 == 128 && g.K % 128 == 0) return launch_bk<T, TC, TA, TB, 32, 32, 128, 2>(g, st);
          if (bk3232 == 256 && g.K % 256 == 0) return launch_bk<T, TC, TA, TB, 32, 32, 256, 1>(g, st);
          if (bk3232 == 257 && g.K % 256 == 0) return launch_bk<T, TC, TA, TB, 32, 32, 256, 2>(g, st);
          return launch_bk<T, TC, TA, TB, 32, 32, 64, 4>(g, st);
        }
        return launch_bk<T, TC, TA, TB, 32, 64, 64, 4>(g, st);
      }
    }
    return launch_one<T, TC, TA, TB, 64, 64>(g, st);
  }
  // float outputs (weight gradients the skinny and the big-tile kernels do not take: the 3072 x 128 feed-forward layers of level 0 at
  // 16 images per GPU, K = 784): 24 tiles of 128 x 128 leave most CUs idle.  Measured per launch (round 6, tools/ab_r06/gpu_r06_f32q.sh):
  // 128 x 64 tiles 27.1 - 28.7 us, 64 x 64 20.4, 32 x 64 17.7 - 18.2, the skinny kernel with K slices 17.8 - 19.7 + its reduce; the step
  // at 16 images +0.4 % (Base), +0.35 % (Large).  VU_GEMM_F32_QUARTER=0: off; =1: 64 x 64; default 32 x 64
  if constexpr (sizeof(TC) == 4 && sizeof(T) == 2) {
    static const int f32q = [] { const char* e = getenv("VU_GEMM_F32_QUARTER"); return e ? atoi(e) : 32; }();
    if (f32q && g.K >= 256 && (long long)vu_cdiv(g.M, 128) * vu_cdiv(g.N, 128) * g.Z1 * g.Z2 < 64 && g.N >= 128 && g.M >= 128)
      return f32q == 32 ? launch_bk<T, TC, TA, TB, 32, 64, 64, 4>(g, st) : launch_one<T, TC, TA, TB, 64, 64>(g, st);
  }
  // between one and ~1.5 big tiles per CU (the level-0 / level-1 linears at 16 - 32 images per GPU: 150 - 312 tiles): half-width
  // tiles balance the chip (measured: M1568 N3072 K3072 107 -> 85 us, M3136 N768 K768 26 -> 20 us; tools/gemm_small_batch.py)
  if constexpr (sizeof(T) == 2) {
    static const int half_below = getenv("VU_GEMM_HALF_BELOW") ? atoi(getenv("VU_GEMM_HALF_BELOW")) : 400;      // measurement switch
    if ((long long)vu_cdiv(g.M, 128) * vu_cdiv(g.N, 128) * g.Z1 * g.Z2 < half_below && g.N >= 128) return launch_one<T, TC, TA, TB, 128, 64>(g, st);
  }
  return launch_one<T, TC, TA, TB, 128, 128>(g, st);
}

template <typename T>
static int launch_layout(vu_gemm_args& g0, int c_float, hipStream_t st) {
  vu_gemm_args g = g0;
  g.swap = 0; g.vecC = 0; g.vec8 = 0;
  const size_t csz = (sizeof(T) == 2 && !c_float) ? 2 : 4;
  // C^T = B^T A^T: same arithmetic, but each lane then owns 4 consecutive elements of a C row,
  // so the epilogue stores (and reads aux / addend) as vectors.  Needs >= 64 original columns.
  // split-K weight-gradient GEMMs add with float atomics: keep their 64-byte-contiguous plain pattern
  const bool will_split = g0.accumulate && (long long)vu_cdiv(g0.M, 128) * vu_cdiv(g0.N, 128) * g0.Z1 * g0.Z2 < 512;
  if (g0.N >= 64 && !will_split) {
    g.A = g0.B; g.B = g0.A; g.M = g0.N; g.N = g0.M;
    g.sAm = g0.sBn; g.sAk = g0.sBk; g.sBk = g0.sAk; g.sBn = g0.sAm;
    g.sA1 = g0.sB1; g.sA2 = g0.sB2; g.sB1 = g0.sA1; g.sB2 = g0.sA2;
    g.swap = 1;
    if (g0.colsum_side) g.colsum_side = 3 - g0.colsum_side;   // the operands changed places
    g.vecC = ((uintptr_t)g0.C % (4 * csz) == 0) && (g0.ldc % 4 == 0) && (g0.sC1 % 4 == 0) && (g0.sC2 % 4 == 0) &&
             (!g0.aux || (uintptr_t)g0.aux % 8 == 0) && (!g0.addend || (uintptr_t)g0.addend % 8 == 0);
  }
  // (original C geometry is the same with or without the operand exchange)
  g.vec8 = (sizeof(T) == 2 && !c_float && g0.N % 8 == 0 && g0.ldc % 8 == 0 && g0.sC1 % 8 == 0 && g0.sC2 % 8 == 0 &&
            (uintptr_t)g0.C % 16 == 0 && (!g0.aux || (uintptr_t)g0.aux % 16 == 0) && (!g0.addend || (uintptr_t)g0.addend % 16 == 0) &&
            (!g0.bias || (uintptr_t)g0.bias % 16 == 0)) ? 1 : 0;
  const bool TA = g.sAk != 1, TB = g.sBk != 1;
  if (TA && g.sAm != 1) { vu_set_error("vu_gemm: A must have a unit stride"); return VU_EINVAL; }
  if (TB && g.sBn != 1) { vu_set_error("vu_gemm: B must have a unit stride"); return VU_EINVAL; }
  constexpr long long VEC = 16 / sizeof(T);
  auto al = [&](const void* p, long long s1, long long s2, long long big) {
    return ((uintptr_t)p % 16 == 0) && (s1 % VEC == 0) && (s2 % VEC == 0) && (big % VEC == 0);
  };
  g.vecA = al(g.A, g.sA1, g.sA2, TA ? g.sAk : g.sAm) ? 1 : 0;
  g.vecB = al(g.B, g.sB1, g.sB2, TB ? g.sBk : g.sBn) ? 1 : 0;
  if (sizeof(T) == 2 && c_float) {
    if (TA && TB) return launch_tiles<T, float, true, true>(g, st);
    if (!TA && !TB) return launch_tiles<T, float, false, false>(g, st);
    vu_set_error("vu_gemm: float output only for NN / TT operand forms");
    return VU_EUNSUPPORTED;
  }
  if (sizeof(T) == 2 && g.accumulate) { vu_set_error("vu_gemm: accumulate needs float C"); return VU_EINVAL; }
  if (!TA && !TB) return launch_tiles<T, T, false, false>(g, st);
  if (!TA && TB) return launch_tiles<T, T, false, true>(g, st);
  if (TA && !TB) return launch_tiles<T, T, true, false>(g, st);
  return launch_tiles<T, T, true, true>(g, st);
}

namespace { thread_local void* g_scratch = nullptr; thread_local size_t g_scratch_bytes = 0; }
void vu_gemm_set_scratch(void* p, size_t bytes) { g_scratch = p; g_scratch_bytes = p ? bytes : 0; }
void vu_gemm_get_scratch(void** p, size_t* bytes) { *p = g_scratch; *bytes = g_scratch_bytes; }

int vu_tsgemm_try(const vu_gemm_args& g, hipStream_t st);      // vu_tsgemm.hip: long-K, small-output weight gradients
int vu_pgemm_try(const vu_gemm_args& g, hipStream_t st);       // vu_pgemm.hip: many rows, small resident weight
int vu_bgemm_try(const vu_gemm_args& g, int c_float, hipStream_t st);   // vu_bgemm.hip: plain big products, one workgroup per CU (LDS-DMA ring)

// dtype: 0 = fp32 storage, 1 = bf16 storage.
int vu_gemm_launch(int dtype, int c_float, vu_gemm_args g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0 || g.Z1 * g.Z2 <= 0) return VU_OK;
  if (g.K <= 0) { vu_set_error("vu_gemm: K must be positive"); return VU_EINVAL; }
  if (dtype == 1) {       // plain big products of levels 0 / 1 (vu_bgemm.hip)
    const int rc = vu_bgemm_try(g, c_float, st);
    if (rc < 0) return rc;
    if (rc > 0) return VU_OK;
  }
  if (dtype == 1 && c_float) {
    const int rc = vu_tsgemm_try(g, st);
    if (rc < 0) return rc;
    if (rc > 0) return VU_OK;
  }
  if (dtype == 1 && !c_float) {
    const int rc = vu_pgemm_try(g, st);
    if (rc < 0) return rc;
    if (rc > 0) return VU_OK;
  }
  if (dtype == 0) {
    // for fp32 storage every C is float; c_float only selects accumulate-capable paths
    return launch_layout<float>(g, 0, st);
  }
  return launch_layout<bf16_t>(g, c_float, st);
}
