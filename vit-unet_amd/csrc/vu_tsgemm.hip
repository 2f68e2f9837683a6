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

// TM x TN: 16 x 16 MFMA tiles per wave; WM x WN waves; PD: k-steps of global loads kept in flight per thread (register ring).
// Bytes in flight are what sets the rate of a stream that only a few workgroups read (Little: ~2 us of loaded-chip latency
// x 45 GB/s per CU = 90 KB per CU): many waves with a short register ring each, rather than few waves with long ones.
template <int TM, int TN, int WM, int WN, int PD>
__global__ __launch_bounds__(64 * WM * WN) void vu_tsgemm_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bm,
                                                                float* __restrict__ C, float* __restrict__ colsum, int M, int N, int K,
                                                                int lda, int ldb, int ldc, int rows_per_split, int nsplit,
                                                                float* __restrict__ slab, float* __restrict__ slab_cs, long long slab_tile) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr int MT = 16 * TM * WM, NT = 16 * TN * WN, BK = 32, PAD = 16;
  constexpr int LDA = MT + PAD, LDB = NT + PAD;
  constexpr int VA = BK * MT / 8, VB = BK * NT / 8;                     // 16-byte vectors per k-step
  constexpr int NV = (VA + VB + NTHR - 1) / NTHR;
  constexpr int OPND = 2 * BK * (LDA + LDB), CST = WM * 16 * (NT + 4) * 2;       // operand buffers / fp32 C staging tile, in bf16 elements
  __shared__ __attribute__((aligned(16))) bf16_t smem[OPND > CST ? OPND : CST];
  bf16_t* As = smem;
  bf16_t* Bs = smem + 2 * BK * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int m_base = blockIdx.y * MT, n_base = blockIdx.z * NT;
  const int k_begin = blockIdx.x * rows_per_split;
  const int k_end = min(K, k_begin + rows_per_split);
  const int nsteps = (k_end - k_begin + BK - 1) / BK;
  uint4 ring[PD][NV];
  // vector v of a k-step: v < VA: A row kk, columns c..c+7; else the same of B
  auto fetch = [&](int step, uint4 (&r)[NV]) {
    const int k0 = k_begin + step * BK;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * NTHR;
      const bool isA = v < VA;
      const int u = isA ? v : v - VA;
      const int per = (isA ? MT : NT) / 8;
      const int kk = u / per, c = (u - kk * per) * 8;
      const bool ok = v < VA + VB && k0 + kk < k_end && (isA ? m_base + c + 8 <= M : n_base + c + 8 <= N);
      // clamped address, zeroed value: the load is unconditional (no branch around it)
      const bf16_t* src = isA ? A + (ok ? (long long)(k0 + kk) * lda + m_base + c : 0) : Bm + (ok ? (long long)(k0 + kk) * ldb + n_base + c : 0);
      const uint4 x = *reinterpret_cast<const uint4*>(src);
      r[i] = ok ? x : make_uint4(0, 0, 0, 0);
    }
  };
  auto commit = [&](int buf, const uint4 (&r)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * NTHR;
      if (v < VA) { const int kk = v / (MT / 8), c = (v - kk * (MT / 8)) * 8; *reinterpret_cast<uint4*>(&As[(buf * BK + kk) * LDA + c]) = r[i]; }
      else if (v < VA + VB) { const int u = v - VA, kk = u / (NT / 8), c = (u - kk * (NT / 8)) * 8; *reinterpret_cast<uint4*>(&Bs[(buf * BK + kk) * LDB + c]) = r[i]; }
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
#pragma unroll
  for (int s_ = 0; s_ < PD; ++s_) fetch(s_ < nsteps ? s_ : 0, ring[s_]);
  const int q = l15 >> 2, p = l15 & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  for (int t0 = 0; t0 < nsteps; t0 += PD) {
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_) {
      const int t = t0 + s_;
      if (t < nsteps) {                                      // workgroup-uniform
        const int buf = t & 1;
        commit(buf, ring[s_]);                               // (buffer buf was last read in step t - 2: every wave has passed the barrier of step t - 1 since)
        fetch(t + PD < nsteps ? t + PD : t, ring[s_]);       // PD - 1 further steps stay in flight during the products
        __syncthreads();
        bf16x8 af[TM], bfr[TN];
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
      }
    }
  }
  // epilogue.  A 16x16 accumulator as it stands is 4 x 64-byte segments per atomic wave-instruction, which the memory-side
  // adders take at a quarter of their rate (measured: the atomics cost 12 - 22 us per launch, more than the k-loop).  So the
  // tile goes through LDS (the operand buffers, 16 rows per wave at a time) and leaves as whole rows: 64 consecutive
  // floats (or two rows of 32) per wave-instruction.
  const bool atomic = nsplit > 1;
  if (nsplit < 0) return;      // (timing experiment: VU_TSGEMM_NOEPI=1 passes a negative split count)
  constexpr int LDC = NT + 4;
  float* Ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();                                        // the k-loop (or the previous pass) is done with the buffer
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(wm * 16 + 4 * lg + r) * LDC + wn * 16 * TN + 16 * j + l15] = acc[i][j][r];
    __syncthreads();
    if (slab) {     // this K slice's partial tile, plain stores: slab[split][mt MT][nt NT], summed in split order by tsgemm_reduce_kernel
      const int Np = gridDim.z * NT;
      float* sp = slab + (long long)blockIdx.x * slab_tile;
      for (int e = tid; e < WM * 16 * NT; e += NTHR) {
        const int lr = e / NT, c = e - lr * NT;
        const int m = m_base + (lr >> 4) * 16 * TM + 16 * i + (lr & 15);
        sp[(long long)m * Np + n_base + c] = Ct[lr * LDC + c];
      }
      if (do_cs && l15 == 0) {
        const int m0 = m_base + wm * 16 * TM + 16 * i + 4 * lg;
#pragma unroll
        for (int r = 0; r < 4; ++r) slab_cs[(long long)blockIdx.x * (gridDim.y * MT) + m0 + r] = cs[i][r];
      }
      continue;
    }
    for (int e = tid; e < WM * 16 * NT; e += NTHR) {
      const int lr = e / NT, c = e - lr * NT;
      const int m = m_base + (lr >> 4) * 16 * TM + 16 * i + (lr & 15), n = n_base + c;
      if (m < M && n < N) {
        float* cp = C + (long long)m * ldc + n;
        const float v = Ct[lr * LDC + c];
        if (atomic) atomicAdd(cp, v); else *cp += v;
      }
    }
    if (do_cs && l15 == 0) {
      const int m0 = m_base + wm * 16 * TM + 16 * i + 4 * lg;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m0 + r < M) { if (atomic) atomicAdd(colsum + m0 + r, cs[i][r]); else colsum[m0 + r] += cs[i][r]; }
    }
  }
}

// C[m][n] += sum over the K slices of slab[s][m][n], and the same for the column sums, in a fixed order (deterministic): one
// float4 item per thread, the slices added in order (interleaved chains, then a fixed tree).  `tile`: floats between two
// slices' tiles (padded so that the slices do not all start on the same memory channels).
#define VU_TS_RED_ITEMS 256      /* float4 items per reduce workgroup: one per thread */
#ifndef VU_TS_RED_CHAINS
#define VU_TS_RED_CHAINS 8
#endif
struct RedDesc {
  const float* slab; const float* slab_cs; float* C; float* colsum;
  int M, N, Mp, Np, ldc, nsplit, nitems, blk0;       // blk0: first workgroup of this reduction inside a batched launch
  long long tile;
};
__device__ __forceinline__ void tsgemm_reduce_body(const float* __restrict__ slab, const float* __restrict__ slab_cs, float* __restrict__ C,
                                                   float* __restrict__ colsum, int M, int N, int Mp, int Np, int ldc, int nsplit, long long tile,
                                                   int nitems, int blk) {
  // (round 5: one float4 item per thread, the slices of an item added in order by four independent chains over loads issued
  // together - 256 consecutive items of a slice are one 4 KB run.  The 16 items x 16 slice lanes form with its LDS exchange ran
  // at 1.3 TB/s: 87 us for the one batched launch of a Base backward at 16 images.)
  const int id = blk * VU_TS_RED_ITEMS + threadIdx.x;
  const int nq = N >> 2;                                    // N % 8 == 0 (checked by the launcher)
  const bool is_c = id < nitems;
  const int ncs = (M + 3) >> 2;                             // column-sum items (float4 of 4 rows' sums; Mp % 4 == 0)
  const bool is_cs = !is_c && colsum && id - nitems < ncs;
  if (!is_c && !is_cs) return;
  int m, n = 0;
  const float* p;
  long long stride;
  if (is_c) { m = id / nq; n = (id - m * nq) * 4; p = slab + (long long)m * Np + n; stride = tile; }
  else { m = (id - nitems) * 4; p = slab_cs + m; stride = Mp; }
  typedef __attribute__((ext_vector_type(4))) float f4v;
  constexpr int CH = VU_TS_RED_CHAINS;
  f4v acc[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) acc[u] = f4v{0.f, 0.f, 0.f, 0.f};
  int s_ = 0;
  for (; s_ + CH - 1 < nsplit; s_ += CH) {
    f4v x[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) x[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p + (long long)(s_ + u) * stride));      // (read once)
#pragma unroll
    for (int u = 0; u < CH; ++u) acc[u] += x[u];
  }
  for (; s_ < nsplit; ++s_) acc[0] += __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p + (long long)s_ * stride));
#pragma unroll
  for (int w = CH / 2; w >= 1; w >>= 1)
#pragma unroll
    for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
  float4 t;
  t.x = acc[0][0]; t.y = acc[0][1]; t.z = acc[0][2]; t.w = acc[0][3];
  if (is_c) {
    float4* cp = reinterpret_cast<float4*>(C + (long long)m * ldc + n);
    float4 c4 = *cp;
    c4.x += t.x; c4.y += t.y; c4.z += t.z; c4.w += t.w;
    *cp = c4;
  } else {
    const float tv[4] = {t.x, t.y, t.z, t.w};
    for (int r = 0; r < 4; ++r) if (m + r < M) colsum[m + r] += tv[r];
  }
}

__global__ __launch_bounds__(256) void tsgemm_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ slab_cs,
                                                            float* __restrict__ C, float* __restrict__ colsum, int M, int N, int Mp,
                                                            int Np, int ldc, int nsplit, long long tile, int nitems) {
  tsgemm_reduce_body(slab, slab_cs, C, colsum, M, N, Mp, Np, ldc, nsplit, tile, nitems, blockIdx.x);
}
// The reductions of SEVERAL products in one launch (their slabs sit side by side in the deferred arena lent by the model
// executor): a workgroup finds its reduction by its index.  29 reduce launches per Base backward were 0.28 ms of tiny
// dependent launches (~10 us each for ~3 us of work); batched they are one launch per backward call.
constexpr int RED_MAX = 40;
struct RedBatch { int n; RedDesc d[RED_MAX]; };
__global__ __launch_bounds__(256) void tsgemm_reduce_batch_kernel(const RedBatch b) {
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.d[i + 1].blk0) ++i;          // (workgroup-uniform)
  const RedDesc& r = b.d[i];
  tsgemm_reduce_body(r.slab, r.slab_cs, r.C, r.colsum, r.M, r.N, r.Mp, r.Np, r.ldc, r.nsplit, r.tile, r.nitems, (int)blockIdx.x - r.blk0);
}

// ---- deferred reductions -------------------------------------------------------------------------------------------------
// vu_tsgemm_set_arena lends a region for the K-slice partial tiles of MANY products; each product takes the next piece of it
// and queues its reduction; vu_tsgemm_flush launches them all (the model executor: once per backward call, before anything
// outside the call reads the gradients).  A product that does not fit flushes first.  Per-thread state, like the scratch.
// Queued reductions of other kernels' per-workgroup partials (vu_gemm.h: vu_defred): one wave per output element, lane l adds the
// partial rows l, l + 64, ... in order over four chains whose loads are issued together, then a fixed shuffle tree - exactly the
// order of the stand-alone reduce kernels (conv_tzw_reduce_kernel, conv_wgrad_mm_reduce_kernel, conv_wgrad_reduce_kernel,
// map_bwd_partials_reduce_kernel).
constexpr int DEF_MAX = 40;      // (40 x 80 bytes of kernel arguments)
struct DefBatch { int n; vu_defred d[DEF_MAX]; };
__global__ __launch_bounds__(1024) void deferred_reduce_batch_kernel(const DefBatch b) {
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.d[i + 1].blk0) ++i;          // (workgroup-uniform)
  const vu_defred& r = b.d[i];
  const int o = ((int)blockIdx.x - r.blk0) * 16 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const float* src = nullptr; float* dst = nullptr;
  long long stride = 0; int cnt = r.nblocks, role = -1;
  if (r.kind == VU_DEFRED_TZW) {
    if (o >= 243) return;
    role = o / 81; const int e = o - role * 81;
    src = r.part + e; stride = 96; cnt = r.nblocks / 3; dst = r.dst[role] + e;
  } else if (r.kind == VU_DEFRED_WGRAD_MM) {
    const int which = o >> 9, tt = (o >> 8) & 1, e = o & 255, n = e >> 4, t = tt * 16 + (e & 15);
    if (o >= 1024 || t >= 27 || n >= (which == 0 ? 3 : 6)) return;
    src = r.part + o; stride = 1024;
    dst = which == 0 ? r.dst[0] + n * 27 + t : (n < 3 ? r.dst[1] + n * 27 + t : r.dst[2] + (n - 3) * 27 + t);
  } else if (r.kind == VU_DEFRED_WGRAD3) {
    const int y = o >> 5, e = o & 31;
    if (y >= r.nconv * 3 || e > 27) return;
    const int cv = y / 3, co = y - cv * 3;
    src = r.part + (long long)y * r.nblocks * 32 + e; stride = 32;
    if (e < 27) dst = r.dst[cv] + co * 27 + e; else if (r.dst[3 + cv]) dst = r.dst[3 + cv] + co; else return;
  } else {
    if (o >= r.n) return;
    src = r.part + o; stride = r.n; dst = o < r.hh ? r.dst[0] + o : r.dst[1] + (o - r.hh);
  }
  // (TZW: the role's workgroups are b = 24 i + 8 role + x, x = 0..7 - the XCD-aware placement of conv_tzw_kernel)
  auto row = [&](int n) -> long long { return role >= 0 ? (long long)(24 * (n >> 3) + 8 * role + (n & 7)) : (long long)n; };
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int n = lane;
  for (; n + 3 * 64 < cnt; n += 4 * 64) {
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = src[row(n + 64 * u) * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += x[u];
  }
  for (int u = 0; n < cnt; n += 64, ++u) s[role >= 0 ? u : 0] += src[row(n) * stride];      // (the tails as the stand-alone kernels have them)
  float a = (s[0] + s[1]) + (s[2] + s[3]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
  if (lane == 0) *dst += a;
}

struct Arena { char* base = nullptr; size_t bytes = 0, off = 0; RedBatch batch; int blocks = 0; DefBatch def; int def_blocks = 0; };
thread_local Arena g_arena;
int arena_flush(hipStream_t st) {
  Arena& a = g_arena;
  int rc = VU_OK;
  if (a.batch.n) {
    hipLaunchKernelGGL(tsgemm_reduce_batch_kernel, dim3((unsigned)a.blocks), dim3(256), 0, st, a.batch);
    if (vu_prof_on()) {
      double bytes = 0.0;
      for (int i = 0; i < a.batch.n; ++i) bytes += (double)(a.batch.d[i].nsplit + 2) * a.batch.d[i].M * a.batch.d[i].N * 4.0;
      vu_prof_note("tsgemm_reduce_batch_kernel", 0.0, bytes);
    }
    rc = vu_check_launch("vu_tsgemm_reduce (batched)");
  }
  if (a.def.n && rc == VU_OK) {
    hipLaunchKernelGGL(deferred_reduce_batch_kernel, dim3((unsigned)a.def_blocks), dim3(1024), 0, st, a.def);
    if (vu_prof_on()) vu_prof_note("deferred_reduce_batch_kernel", 0.0, 0.0);
    rc = vu_check_launch("vu_deferred_reduce (batched)");
  }
  a.batch.n = 0; a.blocks = 0; a.def.n = 0; a.def_blocks = 0; a.off = 0;
  return rc;
}

inline bool ldc_ok(const vu_gemm_args& g) { return g.ldc % 4 == 0 && ((uintptr_t)g.C & 15) == 0; }

template <int TM, int TN, int WM, int WN, int PD>
int launch_ts(const vu_gemm_args& g, hipStream_t st) {
  constexpr int MT = 16 * TM * WM, NT = 16 * TN * WN;
  const int mt = (g.M + MT - 1) / MT, nt = (g.N + NT - 1) / NT;
  const double in_bytes = (double)g.K * (g.M + g.N) * 2.0, out_bytes = (double)g.M * g.N * 4.0;
  void* scr = nullptr; size_t scr_bytes = 0;
  vu_gemm_get_scratch(&scr, &scr_bytes);
  static const bool slab_off = [] { const char* e = getenv("VU_TSGEMM_SLAB"); return e && e[0] == '0'; }();      // A/B switch
  const long long slab_tile = (long long)mt * MT * ((long long)nt * NT) + 1088;        // + 4352 B: spreads the slices over the memory channels
  const size_t per_split = ((size_t)slab_tile + (size_t)mt * MT) * 4;
  Arena& ar = g_arena;
  const bool arena_ok = ar.base && !slab_off && ar.bytes >= 2 * per_split && ldc_ok(g);
  if (arena_ok) { scr = ar.base; scr_bytes = ar.bytes; }          // (sized against the whole arena: a product that does not fit what is left flushes first)
  const bool use_slab = scr && !slab_off && scr_bytes >= 2 * per_split && ldc_ok(g);
  // K slices: enough workgroups to put a stream on every CU and at least 8 k-steps (256 rows) per slice.  With a slab the
  // partial tiles cost a plain write and a read (bounded by half the operand bytes and by the slab); without one they are
  // float atomics on a small, contended output (measured 0.3 TB/s): no more of those than a third of the operand bytes
  static const int want_wgs = [] { const char* e = getenv("VU_TSGEMM_WGS"); return e ? atoi(e) : 256; }();      // measurement switch: workgroups a product aims at
  int want = (want_wgs + mt * nt - 1) / (mt * nt);
  int by_out = (int)(in_bytes / ((use_slab ? 2.0 : 3.0) * out_bytes));
  // (few workgroups even so - the 3072 x 128 outputs of level 0: 16 tiles x 6 slices - : let the slab traffic reach the
  // operand bytes; one CU sustains ~20 GB/s, the stream wants every CU)
  if (use_slab && (long long)by_out * mt * nt < 160) by_out = (int)(in_bytes / out_bytes);
  if (want > by_out) want = by_out;
  const int by_rows = g.K / 256;
  if (want > by_rows) want = by_rows;
  if (use_slab && (size_t)want * per_split > scr_bytes) want = (int)(scr_bytes / per_split);
  { static const int force = getenv("VU_TSGEMM_SPLITS") ? atoi(getenv("VU_TSGEMM_SPLITS")) : 0; if (force > 0 && force < want) want = force; }   // measurement switch
  if (want < 1) want = 1;
  int rows = (g.K + want - 1) / want;
  rows = (rows + 31) / 32 * 32;
  const int nsplit = (g.K + rows - 1) / rows;
  const bool slab = use_slab && nsplit > 1;
  const bool deferred = slab && arena_ok;
  if (deferred) {
    const size_t need = (size_t)nsplit * per_split;
    if (ar.off + need > ar.bytes || ar.batch.n == RED_MAX) { const int rc_ = arena_flush(st); if (rc_) return rc_; }
    scr = ar.base + ar.off;
    ar.off += (need + 255) / 256 * 256;
  }
  float* sl = slab ? (float*)scr : nullptr;
  float* slcs = slab ? sl + (size_t)nsplit * slab_tile : nullptr;
  hipLaunchKernelGGL((vu_tsgemm_kernel<TM, TN, WM, WN, PD>), dim3(nsplit, mt, nt), dim3(64 * WM * WN), 0, st, (const bf16_t*)g.A,
                     (const bf16_t*)g.B, (float*)g.C, g.colsum, g.M, g.N, g.K, (int)g.sAk, (int)g.sBk, (int)g.ldc, rows,
                     getenv("VU_TSGEMM_NOEPI") ? -nsplit : nsplit, sl, slcs, slab_tile);
  if (vu_prof_on()) {
    char tag[64];
    snprintf(tag, sizeof(tag), "vu_tsgemm_kernel<%dx%d>", MT, NT);
    vu_prof_note(tag, 2.0 * g.M * (double)g.N * g.K, in_bytes + out_bytes);
  }
  { const int rc_ = vu_check_launch("vu_tsgemm"); if (rc_) return rc_; }
  if (deferred) {
    const int nitems = g.M * (g.N / 4);
    const long long items = (long long)nitems + (g.colsum ? (g.M + 3) / 4 : 0);
    RedDesc& d = ar.batch.d[ar.batch.n++];
    d.slab = sl; d.slab_cs = slcs; d.C = (float*)g.C; d.colsum = g.colsum; d.M = g.M; d.N = g.N; d.Mp = mt * MT; d.Np = nt * NT;
    d.ldc = (int)g.ldc; d.nsplit = nsplit; d.nitems = nitems; d.blk0 = ar.blocks; d.tile = slab_tile;
    ar.blocks += (int)((items + VU_TS_RED_ITEMS - 1) / VU_TS_RED_ITEMS);
    return VU_OK;
  }
  if (slab) {
    const int nitems = g.M * (g.N / 4);
    const long long items = (long long)nitems + (g.colsum ? (g.M + 3) / 4 : 0);
    hipLaunchKernelGGL(tsgemm_reduce_kernel, dim3((unsigned)((items + VU_TS_RED_ITEMS - 1) / VU_TS_RED_ITEMS)), dim3(256), 0, st, sl, slcs, (float*)g.C, g.colsum, g.M,
                       g.N, mt * MT, nt * NT, (int)g.ldc, nsplit, slab_tile, nitems);
    if (vu_prof_on()) vu_prof_note("tsgemm_reduce_kernel", 0.0, (double)nsplit * out_bytes + 2.0 * out_bytes);
    return vu_check_launch("vu_tsgemm_reduce");
  }
  return VU_OK;
}

inline int side_class(int n) { return n <= 32 ? 32 : (n <= 64 ? 64 : (n <= 128 ? 128 : 192)); }   // workgroup tile side

}  // namespace

void vu_tsgemm_set_arena(void* p, size_t bytes) {
  Arena& a = g_arena;
  a.base = (char*)p; a.bytes = p ? bytes : 0; a.off = 0; a.batch.n = 0; a.blocks = 0; a.def.n = 0; a.def_blocks = 0;
}
int vu_tsgemm_flush(hipStream_t st) { return arena_flush(st); }

// process-level switch (include/vit_unet_amd.h: vu_set_deferred_reductions; initial value VU_DEFER_RED, read once)
static int g_defer_red = [] { const char* e = getenv("VU_DEFER_RED"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" int vu_set_deferred_reductions(int on) {
  if (on < 0 || on > 1) { vu_set_error("vu_set_deferred_reductions: 0 or 1"); return VU_EINVAL; }
  g_defer_red = on;
  return VU_OK;
}
float* vu_deferred_take(size_t floats, hipStream_t st) {
  Arena& a = g_arena;
  const size_t need = (floats * sizeof(float) + 255) / 256 * 256;
  if (!g_defer_red || !a.base || need > a.bytes) return nullptr;
  if (a.off + need > a.bytes || a.def.n == DEF_MAX) { if (arena_flush(st) != VU_OK) return nullptr; }
  float* p = (float*)(a.base + a.off);
  a.off += need;
  return p;
}
void vu_deferred_push(const vu_defred& d) {      // (after a successful vu_deferred_take: there is a free entry)
  Arena& a = g_arena;
  vu_defred& e = a.def.d[a.def.n++];
  e = d;
  e.blk0 = a.def_blocks;
  const int nout = d.kind == VU_DEFRED_TZW ? 243 : (d.kind == VU_DEFRED_WGRAD_MM ? 1024 : (d.kind == VU_DEFRED_WGRAD3 ? d.nconv * 96 : d.n));
  a.def_blocks += (nout + 15) / 16;
}

// 1 = launched, 0 = shape not covered (the caller falls through to the tiled GEMM), < 0 = error
int vu_tsgemm_try(const vu_gemm_args& g, hipStream_t st) {
  static const bool off = [] { const char* e = getenv("VU_TSGEMM"); return e && e[0] == '0'; }();     // A/B switch
  if (off) return 0;
  if (!g.accumulate || g.act || g.dropout || g.addend || g.bias || g.alpha != 1.f || g.Z1 * g.Z2 != 1) return 0;
  if (g.sAm != 1 || g.sBn != 1 || g.colsum_side == 2) return 0;
  if (g.M % 8 || g.N % 8 || g.sAk % 8 || g.sBk % 8 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15)) return 0;
  static const int mink = [] { const char* e = getenv("VU_TSGEMM_MINK"); return e ? atoi(e) : 1024; }();      // measurement switch
  if (g.K < mink) return 0;                                         // long K; at most 16 tiles of 192 x 192 (below)
  if (g.M > 192 && g.N > 192) {
    // 768-class square outputs: no faster than the tiled kernel with float atomics (53 + 7 us against 61), but with a slab
    // the sum is deterministic - taken only when the executor has lent one
    void* scr = nullptr; size_t scr_bytes = 0;
    vu_gemm_get_scratch(&scr, &scr_bytes);
    if (!scr) return 0;
  }
  const int cm = side_class(g.M), cn = side_class(g.N);
  const long long tiles = (long long)((g.M + cm - 1) / cm) * ((g.N + cn - 1) / cn);
  if (tiles > 16) return 0;
#define VU_TS(CM, CN, TMv, TNv, WMv, WNv, PDv) \
  if (cm == CM && cn == CN) { int rc = launch_ts<TMv, TNv, WMv, WNv, PDv>(g, st); return rc ? rc : 1; }
  //    tile       per-wave tiles  waves   ring
  // 192 x 192: two column halves per K slice (192 x 96 tiles): one CU sustains only ~20 GB/s of loads + slab stores, so the
  // product wants every CU streaming, even at 1.5x the operand traffic (A is read by both halves)
  {
    static const bool halves_off = [] { const char* e = getenv("VU_TSGEMM_HALVES"); return e && e[0] == '0'; }();    // A/B switch
    if (cm == 192 && cn == 192 && tiles == 1 && !halves_off) { int rc = launch_ts<3, 3, 4, 2, 4>(g, st); return rc ? rc : 1; }
  }
  VU_TS(192, 192, 3, 6, 4, 2, 4)                                   // (16 waves of 3 x 3 tiles spill at their 128-register cap)
  VU_TS(192, 128, 3, 4, 4, 2, 4) VU_TS(128, 192, 4, 3, 2, 4, 4)
  VU_TS(192, 64, 3, 2, 4, 2, 4)  VU_TS(64, 192, 2, 3, 2, 4, 4)
  VU_TS(192, 32, 3, 1, 4, 2, 4)  VU_TS(32, 192, 1, 3, 2, 4, 4)
  VU_TS(128, 128, 2, 4, 4, 2, 4)
  VU_TS(128, 64, 2, 2, 4, 2, 4)  VU_TS(64, 128, 2, 2, 2, 4, 4)
  VU_TS(128, 32, 2, 1, 4, 2, 4)  VU_TS(32, 128, 1, 2, 2, 4, 4)
  VU_TS(64, 64, 2, 2, 2, 2, 4)   VU_TS(64, 32, 2, 1, 2, 2, 4)  VU_TS(32, 64, 1, 2, 2, 2, 4)
#undef VU_TS
  return 0;
}
