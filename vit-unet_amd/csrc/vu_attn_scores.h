// Fused re-attention kernels for gfx950.
//
// attn_scores_kernel (K6+K7+K8, model.py:155-157): one workgroup = 64 query rows of one (batch,
// head); the head's whole K (N x d, always 2*P/h bytes = 37.6 KB in bf16 for 224x224x3) is staged
// in LDS once, each wave computes its 16 x N logits tile with MFMA into registers (N <= 784:
// 49 accumulator tiles), does the row softmax with 16-lane shuffles, draws the dropout mask from
// the counter hash and writes the sign-tagged probabilities - the (B,h,N,N) logits never touch
// HBM.  Algorithmic traffic: one write of the map (E*|T|) + q, k reads.

// (Template source shared by four translation units - {fp32, bf16} x {softmax, plain product} - so that the
// many register-resident instantiations compile in parallel.)
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "vu_kernels.h"

namespace vu_scores {


template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KS = 32;          // k per MFMA
  static constexpr int FE = 8;           // elements per lane fragment
  typedef bf16x8 Frag;
  static __device__ __forceinline__ Frag zero() { return Frag{0, 0, 0, 0, 0, 0, 0, 0}; }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KS = 4;
  static constexpr int FE = 1;
  typedef float Frag;
  static __device__ __forceinline__ Frag zero() { return 0.f; }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// fragment of a k-contiguous row: FE consecutive elements starting at p (k bounds handled by caller)
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag load_frag(const T* p, int kvalid, bool vec) {
  typedef typename Mma<T>::Frag Frag;
  constexpr int FE = Mma<T>::FE;
  if constexpr (FE == 1) {
    return kvalid > 0 ? (float)p[0] : 0.f;
  } else {
    if (vec && kvalid >= FE) return *reinterpret_cast<const Frag*>(p);
    Frag f = Mma<T>::zero();
#pragma unroll
    for (int e = 0; e < FE; ++e)
      if (e < kvalid) f[e] = p[e];
    return f;
  }
}

// DP = head dim padded to a multiple of 32 ; NT = max 16-column tiles (N <= 16*NT)
template <typename T, int NT, int DP, int WAVES, bool EXACT, bool SOFTMAX>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu((!SOFTMAX && NT > 26) ? 4 : 1, 8))) void attn_scores_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          T* __restrict__ Ps, int N, int D, int H, int d, int ld,
                                                          float scale, vu_rng rng_in) {
  typedef Mma<T> MM;
  typedef typename MM::Frag Frag;
  constexpr int KSTEPS = DP / MM::KS;
  constexpr int LDK = DP + (sizeof(T) == 2 ? 8 : 4);   // LDS row stride (elements)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ks = reinterpret_cast<T*>(smem_raw);
  const vu_rng rng = vu_rng_resolve(rng_in);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int bz = blockIdx.y, b = bz / H, g = bz % H;
  const T* qb = q + (long long)b * N * D + g * d;
  const T* kb = k + (long long)b * N * D + g * d;
  const bool vec = (d % (16 / (int)sizeof(T)) == 0);   // head slices 16-byte aligned

  // ---- stage K_g (N x d, zero-padded to DP columns) in LDS ---------------------------------
  {
    constexpr int VE = 16 / sizeof(T);
    const int chunks_per_row = DP / VE;
    const int total = ((N + 15) & ~15) * chunks_per_row;      // rows N .. 16 ceil(N/16) - 1 are zero: the key tiles read them unconditionally
    for (int c = tid; c < total; c += WAVES * 64) {
      const int row = c / chunks_per_row, kc = (c % chunks_per_row) * VE;
      alignas(16) T tmp[VE];
      if (row >= N) {
#pragma unroll
        for (int e = 0; e < VE; ++e) tmp[e] = (T)0.f;
      } else if (vec && kc + VE <= d) {
        *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(kb + (long long)row * D + kc);
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) tmp[e] = (kc + e < d) ? kb[(long long)row * D + kc + e] : (T)0.f;
      }
      // keys of whole 64-key groups are stored permuted (key 64u + 32 s1 + 8 a + 4 s0 + c -> LDS row
      // 64u + 16 (2 s1 + s0) + 4 a + c): see the tile -> key map below
      const int lrow = row < (EXACT ? ((NT * 16) >> 6) << 6 : (N >> 6) << 6)
                           ? ((row & ~31) | (((row >> 2) & 1) << 4) | (((row >> 3) & 3) << 2) | (row & 3)) : row;
      *reinterpret_cast<uint4*>(&Ks[lrow * LDK + kc]) = *reinterpret_cast<uint4*>(tmp);
    }
  }
  __syncthreads();
  // The workgroup staged K_g once; its waves now walk the 16-row query tiles of this (sample, head)
  // with no further barrier: a wave's stores drain while it multiplies its next tile.
  const int nrt = (N + 15) >> 4;
  // the q fragments of the next row tile are fetched while the current tile is multiplied and stored
  Frag qn[KSTEPS];
  auto load_q = [&](int rt) {
    const int row = rt * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int k0 = ks * MM::KS + lg * MM::FE;
      qn[ks] = (rt < nrt && row < N) ? load_frag<T>(qb + (long long)row * D + k0, d - k0, vec) : MM::zero();
    }
  };
  load_q(blockIdx.x * WAVES + wave);
  for (int rt = blockIdx.x * WAVES + wave; rt < nrt; rt += gridDim.x * WAVES) {
  const int i0 = rt * 16;
  Frag qf[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = qn[ks];
  load_q(rt + gridDim.x * WAVES);

  // Swapped product S^T = K Q^T: the accumulator of tile nt holds, for query i0+l15 (the lane's
  // column), four consecutive keys j0(nt) + r (r = 0..3), so the row softmax reduces in-lane plus
  // two shuffles, a dropout hash word serves an in-lane key pair, and the stores are vectors.
  // Which key an accumulator row stands for is free (it only picks the K row an A-operand lane
  // reads): whole groups of 4 tiles (64 keys) are dealt so that a lane's 4 tiles hold 16
  // CONSECUTIVE keys per tile pair, 64u + 32 (s >> 1) + 8 lg + 4 (s & 1) + r - one 16-byte store per pair, the
  // four lane groups of a row side by side (64 contiguous bytes per instruction), a query row in full 128-byte
  // segments (4 lanes x 32 B) with 16-byte stores instead of 32-byte segments of 8-byte stores.
  // The permutation lives in the K staging pass (LDS row order), so fragment reads stay conflict-free.
  // EXACT: N == 16 * NT - every tile exists and is full, so all tile conditions fold at compile time
  const int ntiles = EXACT ? NT : (N + 15) >> 4;
  const int ngt = EXACT ? ((NT * 16) >> 6) << 2 : (N >> 6) << 2;   // tiles in whole (fully valid) groups of 64 keys
  // (opaque copy of lg: otherwise every per-tile column / hash index is hoisted out of the row-tile loop as a
  // loop invariant, ~60 live registers that spill - and a scratch reload's vmcnt wait drains the store stream)
  int lgv = lg;
  asm volatile("" : "+v"(lgv));
  auto j0_of = [&](int nt) { return nt < ngt ? ((nt >> 2) << 6) + ((nt & 2) << 4) + (lgv << 3) + ((nt & 1) << 2) : nt * 16 + lgv * 4; };
  // accumulators: the softmax form needs the whole row (NT tiles); the plain product has no row-wide dependency and
  // is done in two column halves for long rows - half the accumulator registers, twice the waves per SIMD
  constexpr bool HALVES = !SOFTMAX && NT > 26;
  constexpr int NTA = HALVES ? 26 : NT;            // tiles per pass (even: tile pairs never straddle a pass)
  f32x4 acc[NTA];
  auto compute = [&](int t0) {
#pragma unroll
    for (int nt = 0; nt < NTA; ++nt) {
      acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t0 + nt < ntiles) {
        const int key = (t0 + nt) * 16 + l15;       // LDS row (the staging pass applied the group permutation; rows >= N are zero)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          Frag kf;
          // (a per-lane `key < N` select around the read made every fragment an exec-masked LDS read with its own wait)
          if constexpr (MM::FE == 1) kf = (float)Ks[key * LDK + ks * MM::KS + lg];
          else kf = *reinterpret_cast<const Frag*>(&Ks[key * LDK + ks * MM::KS + lg * MM::FE]);
          acc[nt] = MM::mma(kf, qf[ks], acc[nt]);
        }
      }
      if constexpr (NT > 16) { if (nt % 7 == 6) __builtin_amdgcn_sched_barrier(0); }   // bound the K fragments in flight
    }
  };
  // store tiles nt (and nt+1 when both sit in a whole group: 8 consecutive keys, one 16-byte store for bf16)
  auto store_tiles = [&](T* prow, int nt, bool pair, const vu_f4& oa, const vu_f4& ob) {
    const int j0 = j0_of(nt);
    if constexpr (sizeof(T) == 2) {
      if (pair) {
        union { uint4 u; bf16_t h[8]; } pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pk.h[r] = (bf16_t)oa.v[r]; pk.h[4 + r] = (bf16_t)ob.v[r]; }
        *reinterpret_cast<uint4*>(prow + j0) = pk.u;
        return;
      }
    }
    if (EXACT || j0 < ld) vu_st4(prow + j0, oa);
    if (pair) { const int j1 = j0_of(nt + 1); if (EXACT || j1 < ld) vu_st4(prow + j1, ob); }
  };
  if constexpr (!SOFTMAX) {   // plain product (dAhat = dO v^T in the backward): scaled vector stores
    const int i = i0 + l15;
    T* prow = Ps + ((long long)bz * N + (i < N ? i : 0)) * ld;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += NTA) {
      compute(t0);
      if (i < N) {
#pragma unroll
        for (int nt = 0; nt < NTA; nt += 2) {
          const int gt = t0 + nt;                   // global tile index
          if (gt < ntiles && gt < NT) {
            const bool pair = (nt + 1 < NTA) && (gt + 1 < NT) && (gt + 1 < ntiles);
            vu_f4 oa, ob = {{0.f, 0.f, 0.f, 0.f}};
            const int ja = j0_of(gt);
#pragma unroll
            for (int r = 0; r < 4; ++r) oa.v[r] = (EXACT || ja + r < N) ? acc[nt][r] * scale : 0.f;
            if (nt + 1 < NTA) {
              const int jb = j0_of(gt + 1);
#pragma unroll
              for (int r = 0; r < 4; ++r) ob.v[r] = (EXACT || jb + r < N) ? acc[nt + 1][r] * scale : 0.f;
            }
            if (pair && gt + 1 < ngt) store_tiles(prow, gt, true, oa, ob);
            else { store_tiles(prow, gt, false, oa, oa); if (pair) store_tiles(prow, gt + 1, false, ob, ob); }
          }
        }
      }
    }
    continue;
  }
  compute(0);
  // ---- row softmax (logits rounded to the storage type first, like the unfused path) ------------
  // Only the last key tile can be partial: full tiles take a mask-free path (wave-uniform branch).
  float mx = -INFINITY;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (nt < ntiles) {
      const int j0 = j0_of(nt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sv = acc[nt][r] * scale;
        if constexpr (sizeof(T) == 2) sv = (float)(bf16_t)sv;
        if constexpr (!EXACT) sv = (j0 + r < N) ? sv : -INFINITY;
        acc[nt][r] = sv;
        mx = fmaxf(mx, sv);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
  const float mxl = mx * 1.44269504088896341f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (nt < ntiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e;
        // fp32 storage: subtract first (exact for nearby floats; large logits would lose bits in x*log2e - m*log2e).
        // bf16 storage: the logits were just rounded to 8 significant bits, one fma + exp2 is ample.
        if constexpr (sizeof(T) == 2) e = __builtin_amdgcn_exp2f(fmaf(acc[nt][r], 1.44269504088896341f, -mxl));
        else e = __expf(acc[nt][r] - mx);          // exp(-inf) = 0 for masked keys
        acc[nt][r] = e;
        sum += e;
      }
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  // ---- dropout + sign-tagged store -------------------------------------------------------------
  // mask index of element (row, j) = row * ld + j with ld even: keys (j0, j0+1) and (j0+2, j0+3)
  // each share one 32-bit hash word (16 bits per element).
  const int i = i0 + l15;
  if (i < N) {
    const uint64_t rowi = (uint64_t)bz * N + i;
    T* prow = Ps + rowi * ld;
    const uint32_t ib32 = (uint32_t)(rowi * (uint64_t)ld);   // launcher guarantees < 2^32 map elements
    const uint32_t thr = rng.thr;
    auto tile_out = [&](int nt, const f32x4& a) {
      const int j0 = j0_of(nt);
      float o0 = a[0] * inv, o1 = a[1] * inv, o2 = a[2] * inv, o3 = a[3] * inv;
      if (thr) {
        // 16-bit lanes of two hash words; (x - thr) is negative exactly when x < thr: its sign
        // bit is the "dropped" tag, XOR-ed into the sign of the probability
        const uint32_t wa = vu_hash_word32(rng, ib32 + j0), wb = vu_hash_word32(rng, ib32 + j0 + 2);
        o0 = __uint_as_float(__float_as_uint(o0) ^ (((wa & 0xffffu) - thr) & 0x80000000u));
        o1 = __uint_as_float(__float_as_uint(o1) ^ (((wa >> 16) - thr) & 0x80000000u));
        o2 = __uint_as_float(__float_as_uint(o2) ^ (((wb & 0xffffu) - thr) & 0x80000000u));
        o3 = __uint_as_float(__float_as_uint(o3) ^ (((wb >> 16) - thr) & 0x80000000u));
      }
      if constexpr (!EXACT) {   // zero the padding columns of the partial tile
        if (j0 + 0 >= N) o0 = 0.f;
        if (j0 + 1 >= N) o1 = 0.f;
        if (j0 + 2 >= N) o2 = 0.f;
        if (j0 + 3 >= N) o3 = 0.f;
      }
      const vu_f4 o = {{o0, o1, o2, o3}};
      return o;
    };
#pragma unroll
    for (int nt = 0; nt < NT; nt += 2) {
      if (nt < ntiles) {
        const bool pair = (nt + 1 < NT) && (nt + 1 < ntiles);
        const vu_f4 oa = tile_out(nt, acc[nt]);
        vu_f4 ob = oa;
        if (nt + 1 < NT) { if (pair) ob = tile_out(nt + 1, acc[nt + 1]); }
        if (pair && nt + 1 < ngt) store_tiles(prow, nt, true, oa, ob);
        else { store_tiles(prow, nt, false, oa, oa); if (pair) store_tiles(prow, nt + 1, false, ob, ob); }
      }
      // keep the tiles' hash / tag / store chains apart: scheduled together they need > 256 registers
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }   // row tiles
}

template <typename T, int NT, int DP, int WAVES, bool SOFTMAX>
int launch_scores_w(const T* q, const T* k, T* Ps, int B, int N, int D, int H, int ld, float scale, vu_rng rng,
                    hipStream_t st) {
  constexpr bool softmax = SOFTMAX;
  const int d = D / H;
  constexpr int LDK = DP + (sizeof(T) == 2 ? 8 : 4);
  const size_t lds = (size_t)((N + 15) & ~15) * LDK * sizeof(T);
  const bool full = N == 16 * NT;
  auto kern = full ? attn_scores_kernel<T, NT, DP, WAVES, true, SOFTMAX> : attn_scores_kernel<T, NT, DP, WAVES, false, SOFTMAX>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { vu_set_error("attn_scores: cannot reserve %zu bytes of LDS", lds); return VU_ELAUNCH; }
  }
  // one workgroup per (sample, head) stages K once and walks all row tiles; small batches split the
  // row tiles over several workgroups so that at least ~512 are in flight
  const int nrt = (N + 15) / 16, maxsplit = (nrt + WAVES - 1) / WAVES;
  int nsplit = (512 + B * H - 1) / (B * H);
  if (nsplit > maxsplit) nsplit = maxsplit;
  if (nsplit < 1) nsplit = 1;
  dim3 grid((unsigned)nsplit, (unsigned)(B * H));
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, st, q, k, Ps, N, D, H, d, ld, scale, rng);
  if (vu_prof_on()) vu_prof_note(softmax ? "attn_scores_kernel" : "attn_dscores_kernel", 2.0 * B * H * (double)N * N * d,
                                 ((double)B * H * N * N + 2.0 * B * N * D) * sizeof(T));
  return vu_check_launch("vu_attn_scores");
}

template <typename T, int NT, int DP, bool SOFTMAX>
int launch_scores(const T* q, const T* k, T* Ps, int B, int N, int D, int H, int ld, float scale, vu_rng rng,
                  hipStream_t st) {
  // long rows (N > 208): 128-row workgroups (8 waves) halve the K re-staging; measured 2.0 ms vs
  // 3.25 ms per step against 64-row workgroups on Base (profiles/).
  // long rows: the plain product is store-bound and takes 7 waves (49 row tiles = 7 x 7 at N = 784); the softmax
  // form is VALU-bound (exp + dropout hash, 232 VGPRs = 2 waves per SIMD): 4-wave workgroups, two per CU from
  // different (sample, head) pairs, so that one multiplies while the other stages its K (measured per step at
  // Base: 8 waves x 7 splits 1.53 ms, 4 waves x 2 splits 1.36, 4 waves x 1 split 1.27)
  if constexpr (NT > 13) {
    if constexpr (SOFTMAX) return launch_scores_w<T, NT, DP, 4, true>(q, k, Ps, B, N, D, H, ld, scale, rng, st);
    else return launch_scores_w<T, NT, DP, 7, false>(q, k, Ps, B, N, D, H, ld, scale, rng, st);
  }
  else if constexpr (NT == 13) {
    // 9..13 row tiles (level 1, N = 196): 7 waves walk them in two rounds instead of four
    if (N > 128) return launch_scores_w<T, NT, DP, 7, SOFTMAX>(q, k, Ps, B, N, D, H, ld, scale, rng, st);
    return launch_scores_w<T, NT, DP, 4, SOFTMAX>(q, k, Ps, B, N, D, H, ld, scale, rng, st);
  }
  else return launch_scores_w<T, NT, DP, 4, SOFTMAX>(q, k, Ps, B, N, D, H, ld, scale, rng, st);
}

// ---------------------------------------------------------------------------------------------
// attn_scores_long_kernel: rows too long for the register-resident form (N > 784, d <= 32:
// Lite level 2 has N = 3136, the 512x512 config N = 4096).  K streams through LDS in chunks of
// 512 keys; with softmax the kernel sweeps the keys twice - sweep 1 keeps an online (max, sum)
// per lane, sweep 2 recomputes the logits with MFMA (K = d is tiny) and writes the tagged
// probabilities - so the logits still never reach HBM.
// ---------------------------------------------------------------------------------------------
template <typename T, bool EXACT, bool SOFTMAX>
__global__ __launch_bounds__(512) void attn_scores_long_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               T* __restrict__ Ps, int N, int D, int H, int d, int ld,
                                                               float scale, vu_rng rng_in) {
  typedef Mma<T> MM;
  typedef typename MM::Frag Frag;
  constexpr int DP = 32, CH = 512, WAVES = 8;
  constexpr int KSTEPS = DP / MM::KS;
  constexpr int LDK = DP + (sizeof(T) == 2 ? 8 : 4);
  __shared__ __attribute__((aligned(16))) T Ks[CH * LDK];
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int bz = blockIdx.y, b = bz / H, g = bz % H;
  const T* qb = q + (long long)b * N * D + g * d;
  const T* kb = k + (long long)b * N * D + g * d;
  const bool vec = (d % (16 / (int)sizeof(T)) == 0);
  const int i0 = blockIdx.x * (WAVES * 16) + wave * 16;
  Frag qf[KSTEPS];
  {
    const int row = i0 + l15;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int k0 = ks * MM::KS + lg * MM::FE;
      qf[ks] = (row < N) ? load_frag<T>(qb + (long long)row * D + k0, d - k0, vec) : MM::zero();
    }
  }
  const int i = i0 + l15;
  const uint64_t rowi = (uint64_t)bz * N + (i < N ? i : 0);
  T* prow = Ps + rowi * ld;
  const uint32_t ib32 = (uint32_t)(rowi * (uint64_t)ld);
  const uint32_t thr = rng.thr;
  float mrun = -INFINITY, srun = 0.f, Mx = 0.f, inv = 0.f;
  for (int sweep = 0; sweep < (SOFTMAX ? 2 : 1); ++sweep) {
    for (int c0 = 0; c0 < N; c0 += CH) {
      __syncthreads();
      {  // stage keys [c0, c0+CH) (zero-padded to DP columns)
        constexpr int VE = 16 / sizeof(T);
        constexpr int cpr = DP / VE;
        for (int c = tid; c < CH * cpr; c += WAVES * 64) {
          const int row = c / cpr, kc = (c % cpr) * VE;
          alignas(16) T tmp[VE];
          const int key = c0 + row;
          if (key < N && vec && kc + VE <= d) {
            *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(kb + (long long)key * D + kc);
          } else {
#pragma unroll
            for (int e = 0; e < VE; ++e) tmp[e] = (key < N && kc + e < d) ? kb[(long long)key * D + kc + e] : (T)0.f;
          }
          *reinterpret_cast<uint4*>(&Ks[row * LDK + kc]) = *reinterpret_cast<uint4*>(tmp);
        }
      }
      __syncthreads();
      const int ntl = (N - c0 < CH ? N - c0 : CH);
      for (int nt = 0; nt * 16 < ntl; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int key = nt * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          Frag kf;
          if constexpr (MM::FE == 1) kf = (float)Ks[key * LDK + ks * MM::KS + lg];
          else kf = *reinterpret_cast<const Frag*>(&Ks[key * LDK + ks * MM::KS + lg * MM::FE]);
          acc = MM::mma(kf, qf[ks], acc);
        }
        const int j0 = c0 + nt * 16 + lg * 4;       // this lane's 4 consecutive keys
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sv = acc[r] * scale;
          if constexpr (SOFTMAX) {
            if constexpr (sizeof(T) == 2) sv = (float)(bf16_t)sv;
            if constexpr (!EXACT) sv = (j0 + r < N) ? sv : -INFINITY;
          }
          x[r] = sv;
        }
        if constexpr (!SOFTMAX) {
          if (i < N && (EXACT || j0 < ld)) {
            vu_f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o.v[r] = (EXACT || j0 + r < N) ? x[r] : 0.f;
            vu_st4(prow + j0, o);
          }
        } else if (sweep == 0) {
          const float tm = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
          const float mn = fmaxf(mrun, tm);
          if (mn > -INFINITY) {   // differences first (exact for nearby floats), then the exponential
            srun = srun * __expf(mrun - mn) + __expf(x[0] - mn) + __expf(x[1] - mn) + __expf(x[2] - mn) + __expf(x[3] - mn);
            mrun = mn;
          }
        } else if (i < N && (EXACT || j0 < ld)) {
          float o0 = __expf(x[0] - Mx) * inv, o1 = __expf(x[1] - Mx) * inv;
          float o2 = __expf(x[2] - Mx) * inv, o3 = __expf(x[3] - Mx) * inv;
          if (thr) {
            const uint32_t wa = vu_hash_word32(rng, ib32 + j0), wb = vu_hash_word32(rng, ib32 + j0 + 2);
            o0 = __uint_as_float(__float_as_uint(o0) ^ (((wa & 0xffffu) - thr) & 0x80000000u));
            o1 = __uint_as_float(__float_as_uint(o1) ^ (((wa >> 16) - thr) & 0x80000000u));
            o2 = __uint_as_float(__float_as_uint(o2) ^ (((wb & 0xffffu) - thr) & 0x80000000u));
            o3 = __uint_as_float(__float_as_uint(o3) ^ (((wb >> 16) - thr) & 0x80000000u));
          }
          if constexpr (!EXACT) {
            if (j0 + 0 >= N) o0 = 0.f;
            if (j0 + 1 >= N) o1 = 0.f;
            if (j0 + 2 >= N) o2 = 0.f;
            if (j0 + 3 >= N) o3 = 0.f;
          }
          vu_f4 o = {{o0, o1, o2, o3}};
          vu_st4(prow + j0, o);
        }
      }
    }
    if (SOFTMAX && sweep == 0) {   // combine the 4 lanes (lg = 0..3) that share a query
      float M = fmaxf(mrun, __shfl_xor(mrun, 16, 64));
      M = fmaxf(M, __shfl_xor(M, 32, 64));
      float sc = (mrun > -INFINITY) ? srun * __expf(mrun - M) : 0.f;
      sc += __shfl_xor(sc, 16, 64);
      sc += __shfl_xor(sc, 32, 64);
      Mx = M;
      inv = 1.0f / sc;
    }
  }
}

template <typename T, bool SOFTMAX>
int launch_scores_long(const void* q, const void* k, void* Ps, int B, int N, int D, int H, int ld, float scale, vu_rng rng,
                       hipStream_t st) {
  constexpr bool softmax = SOFTMAX;
  const int d = D / H;
  auto kern = (N % 16 == 0) ? attn_scores_long_kernel<T, true, SOFTMAX> : attn_scores_long_kernel<T, false, SOFTMAX>;
  dim3 grid((unsigned)((N + 127) / 128), (unsigned)(B * H));
  hipLaunchKernelGGL(kern, grid, dim3(512), 0, st, (const T*)q, (const T*)k, (T*)Ps, N, D, H, d, ld, scale, rng);
  if (vu_prof_on()) vu_prof_note(softmax ? "attn_scores_long_kernel" : "attn_dscores_long_kernel",
                                 (softmax ? 4.0 : 2.0) * B * H * (double)N * N * d, ((double)B * H * N * N + 2.0 * B * N * D) * sizeof(T));
  return vu_check_launch("vu_attn_scores_long");
}

template <typename T, bool SOFTMAX>
int dispatch_scores(const void* q, const void* k, void* Ps, int B, int N, int D, int H, int ld, float scale, vu_rng rng,
                    hipStream_t st) {
  const int d = D / H;
  const int dp = (d + 31) / 32 * 32;
  const int nt = (N + 15) / 16;
#define VU_SC(NTv, DPv) return launch_scores<T, NTv, DPv, SOFTMAX>((const T*)q, (const T*)k, (T*)Ps, B, N, D, H, ld, scale, rng, st)
  if (nt <= 4) {
    if (dp == 32) VU_SC(4, 32); if (dp == 64) VU_SC(4, 64); if (dp == 96) VU_SC(4, 96); if (dp == 128) VU_SC(4, 128);
    if (dp == 192) VU_SC(4, 192); if (dp == 384) VU_SC(4, 384);
  } else if (nt <= 13) {
    if (dp == 32) VU_SC(13, 32); if (dp == 64) VU_SC(13, 64); if (dp == 96) VU_SC(13, 96); if (dp == 128) VU_SC(13, 128);
  } else if (nt <= 49) {
    if (dp == 32) VU_SC(49, 32); if (dp == 64) VU_SC(49, 64);
  } else if (dp == 32) {
    return launch_scores_long<T, SOFTMAX>(q, k, Ps, B, N, D, H, ld, scale, rng, st);
  }
#undef VU_SC
  return 1;   // shape not covered: the caller falls back to GEMM + softmax kernels
}


}  // namespace vu_scores

#define VU_SCORES_TU(name, T, SOFTMAX)                                                                                  \
  int name(const void* q, const void* k, void* Ps, int B, int N, int D, int H, int ld, float scale, vu_rng rng,       \
           hipStream_t st) {                                                                                           \
    return vu_scores::dispatch_scores<T, SOFTMAX>(q, k, Ps, B, N, D, H, ld, scale, rng, st);                           \
  }
