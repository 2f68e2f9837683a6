// Tall-skinny weight-gradient product for bf16 storage:   C[M][N] (fp32) += sum_k A[k][m] * B[k][n],
// K = B*N tokens (3 136 .. 50 176 rows), M x N small (the 192 x 192 / 192 x 32 / 768 x 64 / 3072 x 128 linears of
// model.py:62-80,150-164 seen from the backward).  These products are HBM streams (both operands are read once, the
// output is a few hundred KB), but as tiles of the general GEMM they ran at 0.6 TB/s: with 128 x 128 tiles the output
// has 1 - 24 tiles, so K was split 7 - 98 ways and every block did only ~8 k-steps between its prologue and an epilogue
// of 16 K float atomics.  Here one workgroup owns up to 192 x 192 of the output (4 waves, 2 x 2, up to 6 x 6 MFMA tiles
// each = 144 accumulator registers) and a LONG K slice: the operands are read exactly once when the output fits one
// tile, the number of K slices is bounded by the bytes of float atomics they cost (chip-wide 1.3 TB/s), and the k-loop
// is a one-barrier double-buffered stream (register prefetch of step t+1, LDS image of step t).
// Both operands are k-major ("T" form of vu_gemm.h): LDS images [k][m], fragments by ds_read_b64_tr_b16.
// The bias gradient (column sums over k of A) rides along as one more MFMA column tile against a vector of ones.
#include <stdio.h>
#include <stdlib.h>
#include "vu_gemm.h"

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;

template <int TM, int TN>
__global__ __launch_bounds__(256) void vu_tsgemm_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bm,
                                                        float* __restrict__ C, float* __restrict__ colsum, int M, int N, int K,
                                                        int lda, int ldb, int ldc, int rows_per_split, int nsplit) {
  constexpr int MT = 32 * TM, NT = 32 * TN, BK = 32, PAD = 16;
  constexpr int LDA = MT + PAD, LDB = NT + PAD;
  constexpr int VA = BK * MT / 8, VB = BK * NT / 8;                     // 16-byte vectors per k-step
  constexpr int NVA = (VA + 255) / 256, NVB = (VB + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BK * (LDA + LDB)];
  bf16_t* As = smem;
  bf16_t* Bs = smem + 2 * BK * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m_base = blockIdx.y * MT, n_base = blockIdx.z * NT;
  const int k_begin = blockIdx.x * rows_per_split;
  const int k_end = min(K, k_begin + rows_per_split);
  const int nsteps = (k_end - k_begin + BK - 1) / BK;
  uint4 ra[NVA], rb[NVB];
  auto fetch = [&](int step) {
    const int k0 = k_begin + step * BK;
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int v = tid + i * 256;
      const int kk = v / (MT / 8), c = (v - kk * (MT / 8)) * 8;
      const bool ok = (VA % 256 == 0 || v < VA) && k0 + kk < k_end && m_base + c + 8 <= M;
      // clamped address, zeroed value: the load is unconditional (no branch around it)
      const long long off = ok ? (long long)(k0 + kk) * lda + m_base + c : 0;
      const uint4 x = *reinterpret_cast<const uint4*>(A + off);
      ra[i] = ok ? x : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int v = tid + i * 256;
      const int kk = v / (NT / 8), c = (v - kk * (NT / 8)) * 8;
      const bool ok = (VB % 256 == 0 || v < VB) && k0 + kk < k_end && n_base + c + 8 <= N;
      const long long off = ok ? (long long)(k0 + kk) * ldb + n_base + c : 0;
      const uint4 x = *reinterpret_cast<const uint4*>(Bm + off);
      rb[i] = ok ? x : make_uint4(0, 0, 0, 0);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int v = tid + i * 256;
      if (VA % 256 == 0 || v < VA) { const int kk = v / (MT / 8), c = (v - kk * (MT / 8)) * 8; *reinterpret_cast<uint4*>(&As[(buf * BK + kk) * LDA + c]) = ra[i]; }
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int v = tid + i * 256;
      if (VB % 256 == 0 || v < VB) { const int kk = v / (NT / 8), c = (v - kk * (NT / 8)) * 8; *reinterpret_cast<uint4*>(&Bs[(buf * BK + kk) * LDB + c]) = rb[i]; }
    }
  };
  f32x4 acc[TM][TN], cs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_cs = colsum != nullptr && blockIdx.z == 0 && wn == 0;       // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  if (nsteps > 0) {
    fetch(0);
    commit(0);
  }
  __syncthreads();
  const int q = l15 >> 2, p = l15 & 3;
  for (int t = 0; t < nsteps; ++t) {
    const int buf = t & 1;
    if (t + 1 < nsteps) fetch(t + 1);
    bf16x8 af[TM], bfr[TN];
    typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const bf16_t* a0 = &As[(buf * BK + 8 * lg + q) * LDA + wm * 16 * TM + 16 * i + 4 * p];
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)a0);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a0 + 4 * LDA));
      const s16x8 x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      af[i] = __builtin_bit_cast(bf16x8, x);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bf16_t* b0 = &Bs[(buf * BK + 8 * lg + q) * LDB + wn * 16 * TN + 16 * j + 4 * p];
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)b0);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(b0 + 4 * LDB));
      const s16x8 x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      bfr[j] = __builtin_bit_cast(bf16x8, x);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if (do_cs) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, cs[i], 0, 0, 0);
    }
    if (t + 1 < nsteps) commit(buf ^ 1);
    __syncthreads();
  }
  // epilogue: accumulator rows 4 lg + r = m, column l15 = n
  const bool atomic = nsplit > 1;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m0 = m_base + wm * 16 * TM + 16 * i + 4 * lg;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n_base + wn * 16 * TN + 16 * j + l15;
      if (n < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m0 + r < M) {
            float* cp = C + (long long)(m0 + r) * ldc + n;
            if (atomic) atomicAdd(cp, acc[i][j][r]); else *cp += acc[i][j][r];
          }
      }
    }
    if (do_cs && l15 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m0 + r < M) { if (atomic) atomicAdd(colsum + m0 + r, cs[i][r]); else colsum[m0 + r] += cs[i][r]; }
    }
  }
}

template <int TM, int TN>
int launch_ts(const vu_gemm_args& g, int mt, int nt, hipStream_t st) {
  const double in_bytes = (double)g.K * (g.M + g.N) * 2.0, out_bytes = (double)g.M * g.N * 4.0;
  // K slices: enough blocks to keep the loads of ~200 CUs in flight, but no more float-atomic bytes than a third of the
  // operand bytes, and at least 8 k-steps (256 rows) per slice
  int want = (256 + mt * nt - 1) / (mt * nt);
  const int by_atomics = (int)(in_bytes / (3.0 * out_bytes));
  if (want > by_atomics) want = by_atomics;
  const int by_rows = g.K / 256;
  if (want > by_rows) want = by_rows;
  if (want < 1) want = 1;
  int rows = (g.K + want - 1) / want;
  rows = (rows + 31) / 32 * 32;
  const int nsplit = (g.K + rows - 1) / rows;
  hipLaunchKernelGGL((vu_tsgemm_kernel<TM, TN>), dim3(nsplit, mt, nt), dim3(256), 0, st, (const bf16_t*)g.A, (const bf16_t*)g.B, (float*)g.C,
                     g.colsum, g.M, g.N, g.K, (int)g.sAk, (int)g.sBk, (int)g.ldc, rows, nsplit);
  if (vu_prof_on()) {
    char tag[64];
    snprintf(tag, sizeof(tag), "vu_tsgemm_kernel<%d,%d>", TM, TN);
    vu_prof_note(tag, 2.0 * g.M * (double)g.N * g.K, in_bytes + out_bytes);
  }
  return vu_check_launch("vu_tsgemm");
}

inline int tile_units(int n) { return n <= 32 ? 1 : (n <= 64 ? 2 : (n <= 128 ? 4 : 6)); }   // wave-tile size in 16s, workgroup tile = 32 x that

}  // namespace

// 1 = launched, 0 = shape not covered (the caller falls through to the tiled GEMM), < 0 = error
int vu_tsgemm_try(const vu_gemm_args& g, hipStream_t st) {
  static const bool off = [] { const char* e = getenv("VU_TSGEMM"); return e && e[0] == '0'; }();     // A/B switch
  if (off) return 0;
  if (!g.accumulate || g.act || g.dropout || g.addend || g.bias || g.alpha != 1.f || g.Z1 * g.Z2 != 1) return 0;
  if (g.sAm != 1 || g.sBn != 1 || g.colsum_side == 2) return 0;
  if (g.M % 8 || g.N % 8 || g.sAk % 8 || g.sBk % 8 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15)) return 0;
  if (g.K < 2048 || (g.M > 192 && g.N > 192)) return 0;             // long K, at least one small side
  const int tm = tile_units(g.M), tn = tile_units(g.N);
  const int mt = (g.M + 32 * tm - 1) / (32 * tm), nt = (g.N + 32 * tn - 1) / (32 * tn);
  if (mt * nt > 16 || mt > 65535 || nt > 65535) return 0;
#define VU_TS(TMv, TNv) if (tm == TMv && tn == TNv) { int rc = launch_ts<TMv, TNv>(g, mt, nt, st); return rc ? rc : 1; }
  VU_TS(6, 6) VU_TS(6, 1) VU_TS(1, 6) VU_TS(6, 2) VU_TS(2, 6) VU_TS(6, 4) VU_TS(4, 6)
  VU_TS(4, 4) VU_TS(4, 2) VU_TS(2, 4) VU_TS(4, 1) VU_TS(1, 4) VU_TS(2, 2) VU_TS(2, 1) VU_TS(1, 2)
#undef VU_TS
  return 0;
}
