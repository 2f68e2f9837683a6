// Streaming map x head-slice products (rows / cols forms).  (Split from vu_attn.hip.)
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "vu_kernels.h"
// =============================================================================================
// Map x head-slice products for long rows and small head dims (bf16 storage, d <= 32):
//   rows form: out[b,i,g*d+t] = sum_j M[b,g,i,j] X[b,j,g*d+t]      O = Ahat v ;  dq = dS k
//   cols form: out[b,j,g*d+t] = sum_i M[b,g,i,j] X[b,i,g*d+t]      dv = Ahat^T dO ;  dk = dS^T q
// Both stream the (N x N) map of one (sample, head) exactly once, straight from HBM into MFMA
// operand registers (each lane reads 32 contiguous bytes of a map row, a 16-lane group 128 B): the
// map never passes through LDS.  The head slice X_g (N x d, a few tens of KB) is staged once per
// workgroup, transposed, as Xt[t][n], so that its fragments are 16-/8-byte LDS reads.
// The products are computed transposed (out^T = Xt . M^T) so that a lane ends up with 4
// consecutive t of one token: 8-byte stores into the token-major (B,N,D) activation.
// =============================================================================================
namespace {

// Xt[t][n] = X[n][t] for n < N, t < d; zero elsewhere (t < 16 TT, n < ldk).  LDV = ldk + 8: consecutive
// rows start 4 banks apart, 16 lanes reading 16 B of 16 different rows cover all 64 banks once.
template <int TT>
__device__ __forceinline__ void stage_slice_T(bf16_t* Xt, const bf16_t* __restrict__ Xg, int N, int D, int d, int ldk, int LDV,
                                              int tid, int nthr) {
  const bool vec = (d % 8 == 0) && (D % 8 == 0);
  constexpr int CPT = 2 * TT;          // 8-element chunks per token
  const int cells = ldk * CPT;
  for (int c0 = tid; c0 < cells; c0 += 4 * nthr) {      // 4 independent 16-byte loads in flight per thread
    uint4 x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * nthr;
      const int n = c / CPT, t0 = (c % CPT) * 8;
      x[k] = make_uint4(0, 0, 0, 0);
      if (c < cells && n < N && t0 < d) {
        if (vec) x[k] = *reinterpret_cast<const uint4*>(Xg + (long long)n * D + t0);
        else {
          unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (t0 + e < d) w[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, Xg[(long long)n * D + t0 + e]) << (16 * (e & 1));
          x[k] = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * nthr;
      if (c < cells) {
        const int n = c / CPT, t0 = (c % CPT) * 8;
        const unsigned w[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
        unsigned short* dst = reinterpret_cast<unsigned short*>(Xt) + t0 * LDV + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e * LDV] = (unsigned short)(w[e >> 1] >> (16 * (e & 1)));
      }
    }
  }
}

// Wide head slices (TT > 2) are staged as they lie in memory, Xn[n][mp_pitch] (16-byte LDS stores, no scatter: the
// transposed image above costs 8 two-byte LDS stores per 16 bytes loaded and was half of the kernel at N = 196,
// d = 96); the MFMA fragments then come out of transposing LDS reads (ds_read_b64_tr_b16), two per 16x16x32 operand.
// Pitch 16 TT + 8 elements: the 4 rows x 4 chunks a 16-lane group touches fall in distinct banks.
constexpr int mp_pitch(int tt) { return 16 * tt + 8; }
constexpr bool mp_natural(int tt) { return tt > 2; }
template <int TT>
__device__ __forceinline__ void stage_slice_N(bf16_t* Xn, const bf16_t* __restrict__ Xg, int N, int D, int d, int ldk, int tid, int nthr) {
  constexpr int P = mp_pitch(TT);
  const bool vec = (d % 8 == 0) && (D % 8 == 0);
  constexpr int CPT = 2 * TT;          // 8-element chunks per token
  const int cells = ldk * CPT;
  for (int c0 = tid; c0 < cells; c0 += 4 * nthr) {      // 4 independent 16-byte loads in flight per thread
    uint4 x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * nthr;
      const int n = c / CPT, t0 = (c % CPT) * 8;
      x[k] = make_uint4(0, 0, 0, 0);
      if (c < cells && n < N && t0 < d) {
        if (vec) x[k] = *reinterpret_cast<const uint4*>(Xg + (long long)n * D + t0);
        else {
          unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (t0 + e < d) w[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, Xg[(long long)n * D + t0 + e]) << (16 * (e & 1));
          x[k] = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * nthr;
      if (c < cells) *reinterpret_cast<uint4*>(Xn + (c / CPT) * P + (c % CPT) * 8) = x[k];
    }
  }
}
// column sums of a natural image over its ldk tokens, added to colsum_s[16 TT]; `scratch` holds nparts x 16 TT floats.
// Fixed summation order (partials per token residue, then residue by residue): the same bits every run.
template <int TT>
__device__ __forceinline__ void colsum_N(const bf16_t* Xn, float* scratch, float* colsum_s, int ldk, int nparts, int tid) {
  constexpr int P = mp_pitch(TT), CPT = 2 * TT;
  const int chunk = tid % CPT, part = tid / CPT;
  if (part < nparts) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = part; n < ldk; n += nparts) {
      const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(Xn + n * P + 8 * chunk);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += (float)v8[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) scratch[part * 16 * TT + 8 * chunk + e] = a[e];
  }
  __syncthreads();
  if (tid < 16 * TT) {
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += scratch[p * 16 * TT + tid];
    colsum_s[tid] += s;
  }
  __syncthreads();
}
typedef __attribute__((address_space(3))) s16x4* mp_lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short mp_s16x8;
// 16x16x32 operand (rows = 16 features from f0 [the lane's 4 pp already in p], k-slots = 8 consecutive tokens of the
// lane group) out of a row-major image: p points at (first token of the group + qq, f0 + 4 pp)
template <int PITCH>
__device__ __forceinline__ bf16x8 tr_pair(const bf16_t* p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mp_lds_s16x4_ptr)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mp_lds_s16x4_ptr)(p + 4 * PITCH));
  const mp_s16x8 t8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, t8);
}

// 4 consecutive t (t0..t0+3) of token row `orow` (points at the head slice): vector store when whole
__device__ __forceinline__ void store_t4(bf16_t* orow, int t0, int d, bool vec, const f32x4& a) {
  if (t0 >= d) return;
  if (vec && t0 + 4 <= d) {
    const vu_f4 o = {{a[0], a[1], a[2], a[3]}};
    vu_st4(orow + t0, o);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (t0 + r < d) orow[t0 + r] = (bf16_t)a[r];
  }
}

// Both kernels move the map HBM -> registers (a ring a few steps ahead) -> a small wave-private LDS tile ->
// MFMA fragments.  The detour through LDS is what keeps the global loads whole: a wave instruction reads
// 8 full 128-byte row segments (lane = 16 B of a line), where loading in fragment shape (16 lanes = 16
// different rows) costs 8x the cache-line lookups.  A wave only ever touches its own tile, and LDS
// operations of one wave complete in order, so no barrier is involved.
constexpr int MP_LDT = 72;     // tile row stride in elements (144 B: 16 rows x 16 B cover all banks once)

// CHUNKED (rows longer than one LDS image of X: Lite level 2, N = 3136): the contraction index is cut into chunks of
// mp_ch(TT) tokens; a workgroup then owns WAVES row tiles (one per wave, accumulators live across chunks) and re-stages
// the head slice chunk by chunk.
constexpr int mp_ch(int tt) { return tt <= 2 ? 832 : 256; }     // tokens per staged chunk (image = 16 tt rows x chunk)

template <int WAVES, int TT, bool CHUNKED>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void attn_map_rows_kernel(const bf16_t* __restrict__ M, const bf16_t* __restrict__ X,
                                                                 bf16_t* __restrict__ out, const float* __restrict__ sc, const float* __restrict__ kappa,
                                                                 int N, int D, int H, int d, int ld, int mdiv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr bool NAT = mp_natural(TT);
  constexpr int P = mp_pitch(TT);
  const int ldk = CHUNKED ? mp_ch(TT) : ((N + 63) & ~63), LDV = ldk + 8;     // tokens per staged image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  bf16_t* Xt = reinterpret_cast<bf16_t*>(smem_raw);                       // [16 TT][LDV] (rows >= d stay zero), or [ldk][P]
  bf16_t* T0 = Xt + (NAT ? ldk * P : 16 * TT * LDV);
  bf16_t* T = T0 + wave * (16 * MP_LDT);                                  // this wave's [16][MP_LDT] tile
  const int bz = blockIdx.y, b = bz / H, g = bz % H;
  const bf16_t* Xg = X + (long long)b * N * D + g * d;
  // affine form (M is the centred mixed map Ac, the product wanted is with Ahat = sc_g Ac + kappa_g):
  // out = sc_g (Ac . X) + kappa_g * (column sums of X over the tokens), the sums taken from the staged slice
  __shared__ float colsum_s[16 * TT];
  float a_sc = 1.f, a_kp = 0.f;
  if (sc) { a_sc = sc[g / mdiv]; a_kp = kappa[g / mdiv]; }
  if (tid < 16 * TT) colsum_s[tid] = 0.f;
  auto stage = [&](int n0) {       // tokens [n0, n0 + ldk) of the slice -> Xt, column sums accumulated (affine form)
    if constexpr (NAT) {
      stage_slice_N<TT>(Xt, Xg + (long long)n0 * D, min(N - n0, ldk), D, d, ldk, tid, WAVES * 64);
      __syncthreads();
      // the wave tiles are idle while the slice is staged: scratch of the column sums
      if (sc) colsum_N<TT>(Xt, reinterpret_cast<float*>(T0), colsum_s, ldk, min(WAVES * 64 / (2 * TT), WAVES * 16 * MP_LDT / (32 * TT)), tid);
      return;
    }
    stage_slice_T<TT>(Xt, Xg + (long long)n0 * D, min(N - n0, ldk), D, d, ldk, LDV, tid, WAVES * 64);
    __syncthreads();
    if (sc) {
      const int part = tid % 16;                         // 16 threads per feature row
      for (int t = tid / 16; t < 16 * TT; t += WAVES * 4) {
        float acc_s = 0.f;
        for (int n = part * 8; n < ldk; n += 128) {
          const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(Xt + t * LDV + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc_s += (float)v8[e];
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) acc_s += __shfl_xor(acc_s, m, 64);
        if (part == 0) colsum_s[t] += acc_s;
      }
      __syncthreads();
    }
  };
  const bf16_t* Mb = M + (long long)(bz / mdiv) * N * ld;      // (wide heads: mdiv slices of 96 features share the head's map)
  const int nrt = (N + 15) >> 4;
  const bool vec = (d % 4 == 0) && (D % 4 == 0);
  const int lrow = lane >> 3, lch = (lane & 7) * 8;                        // load shape: 8 rows x 8 chunks of 16 B
  const bf16_t* x0 = NAT ? Xt + (16 * lg + (l15 >> 2)) * P + 4 * (l15 & 3) : Xt + l15 * LDV + 16 * lg;
  // one row tile against the staged tokens: map columns cbase + [0, 64 nsteps)
  auto run = [&](int rt, int cbase, int nsteps, f32x4 (&acc)[TT]) {
    // unconditional loads at clamped addresses (rows >= N re-read row N-1, columns >= ld the row's last chunk):
    // what they return is finite map data that meets zeros of Xt or lands in rows that are never stored.
    // (Selecting between a load and a zero makes hipcc select between POINTERS and emit serialized flat loads.)
    const bf16_t* r0 = Mb + (long long)min(rt * 16 + lrow, N - 1) * ld;
    const bf16_t* r1 = Mb + (long long)min(rt * 16 + 8 + lrow, N - 1) * ld;
    // (named registers, not an array: hipcc left an indexed ring in scratch memory)
    uint4 ma0, mb0, ma1, mb1, ma2, mb2, ma3, mb3, ma4, mb4, ma5, mb5;
    auto fetch = [&](int step, uint4& ma, uint4& mb) {
      const int j = min(cbase + step * 64 + lch, ld - 8);
      ma = *reinterpret_cast<const uint4*>(r0 + j);
      mb = *reinterpret_cast<const uint4*>(r1 + j);
    };
    auto put = [&](const uint4& ma, const uint4& mb) {
      *reinterpret_cast<uint4*>(T + lrow * MP_LDT + lch) = ma;
      *reinterpret_cast<uint4*>(T + (8 + lrow) * MP_LDT + lch) = mb;
    };
    auto mult = [&](int sidx) {
      // B operand: lane (l15 = map row, lg): columns 16 lg + [0,8) and + [8,16) of the step: k-slots of two MFMAs
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(T + l15 * MP_LDT + 16 * lg);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(T + l15 * MP_LDT + 16 * lg + 8);
      if constexpr (NAT) {
        const bf16_t* xa = x0 + sidx * 64 * P;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair<P>(xa + 16 * tt), b0, acc[tt], 0, 0, 0);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair<P>(xa + 16 * tt + 8 * P), b1, acc[tt], 0, 0, 0);
        }
      } else {
        const bf16_t* xa = x0 + sidx * 64;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xa + 16 * tt * LDV), b0, acc[tt], 0, 0, 0);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xa + 16 * tt * LDV + 8), b1, acc[tt], 0, 0, 0);
        }
      }
    };
    auto step = [&](int sidx, uint4& ma, uint4& mb) {
      put(ma, mb);                      // the tile leaves the ring registers ...
      fetch(sidx + 6, ma, mb);          // ... which are re-issued at once, 6 steps ahead
      mult(sidx);
    };
    fetch(0, ma0, mb0); fetch(1, ma1, mb1); fetch(2, ma2, mb2); fetch(3, ma3, mb3); fetch(4, ma4, mb4); fetch(5, ma5, mb5);
    // the main loop is branch-free so that hipcc can count its vmcnt waits (a conditional step makes it drain
    // the whole ring, vmcnt(0), once per trip); the last nsteps % 6 steps follow
    const int nfull = nsteps / 6 * 6;
    for (int s0 = 0; s0 < nfull; s0 += 6) {
      step(s0, ma0, mb0); step(s0 + 1, ma1, mb1); step(s0 + 2, ma2, mb2);
      step(s0 + 3, ma3, mb3); step(s0 + 4, ma4, mb4); step(s0 + 5, ma5, mb5);
    }
    if (nfull + 0 < nsteps) { put(ma0, mb0); mult(nfull + 0); }
    if (nfull + 1 < nsteps) { put(ma1, mb1); mult(nfull + 1); }
    if (nfull + 2 < nsteps) { put(ma2, mb2); mult(nfull + 2); }
    if (nfull + 3 < nsteps) { put(ma3, mb3); mult(nfull + 3); }
    if (nfull + 4 < nsteps) { put(ma4, mb4); mult(nfull + 4); }
  };
  // C[row = t = 4 lg + r (+16)][col = token i]
  auto finish = [&](int rt, f32x4 (&acc)[TT]) {
    const int i = rt * 16 + l15;
    if (i < N) {
      bf16_t* orow = out + ((long long)b * N + i) * D + g * d;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        f32x4 o = acc[tt];
        if (sc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaf(a_sc, o[r], a_kp * colsum_s[16 * tt + 4 * lg + r]);
        }
        store_t4(orow, 16 * tt + 4 * lg, d, vec, o);
      }
    }
  };
  if constexpr (!CHUNKED) {
    __syncthreads();
    stage(0);
    __syncthreads();
    for (int rt = blockIdx.x * WAVES + wave; rt < nrt; rt += gridDim.x * WAVES) {
      f32x4 acc[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      run(rt, 0, ldk >> 6, acc);
      finish(rt, acc);
    }
  } else {
    const int rt = blockIdx.x * WAVES + wave;       // one row tile per wave; every wave takes part in the staging
    f32x4 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < N; n0 += mp_ch(TT)) {
      __syncthreads();                               // the previous chunk's fragments have been read
      stage(n0);
      __syncthreads();
      if (rt < nrt) run(rt, n0, (min(N - n0, mp_ch(TT)) + 63) >> 6, acc);
    }
    if (rt < nrt) finish(rt, acc);
  }
}

// cols form: a wave owns a strip of 64 map columns and walks all rows 32 at a time; the contraction runs
// over map rows, so the B operand (k = row, n = column) comes out of the row-major tile through the
// transposing LDS read (ds_read_b64_tr_b16).
template <int WAVES, int TT, bool CHUNKED>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 8))) void attn_map_cols_kernel(const bf16_t* __restrict__ M, const bf16_t* __restrict__ X,
                                                                 bf16_t* __restrict__ out, const float* __restrict__ sc, const float* __restrict__ kappa,
                                                                 int N, int D, int H, int d, int ld, int mdiv) {
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr bool NAT = mp_natural(TT);
  constexpr int P = mp_pitch(TT);
  const int ldk = CHUNKED ? mp_ch(TT) : ((N + 63) & ~63), LDV = ldk + 8;     // tokens (= map rows here) per staged image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  bf16_t* Xt = reinterpret_cast<bf16_t*>(smem_raw);                       // [16 TT][LDV], or [ldk][P]
  bf16_t* T0 = Xt + (NAT ? ldk * P : 16 * TT * LDV);
  bf16_t* T = T0 + wave * (32 * MP_LDT);                                  // this wave's [32][MP_LDT] tile
  const int bz = blockIdx.y, b = bz / H, g = bz % H;
  const bf16_t* Xg = X + (long long)b * N * D + g * d;
  // affine form: see the rows kernel
  __shared__ float colsum_s[16 * TT];
  float a_sc = 1.f, a_kp = 0.f;
  if (sc) { a_sc = sc[g / mdiv]; a_kp = kappa[g / mdiv]; }
  if (tid < 16 * TT) colsum_s[tid] = 0.f;
  auto stage = [&](int n0) {
    if constexpr (NAT) {
      stage_slice_N<TT>(Xt, Xg + (long long)n0 * D, min(N - n0, ldk), D, d, ldk, tid, WAVES * 64);
      __syncthreads();
      if (sc) colsum_N<TT>(Xt, reinterpret_cast<float*>(T0), colsum_s, ldk, min(WAVES * 64 / (2 * TT), WAVES * 32 * MP_LDT / (32 * TT)), tid);
      return;
    }
    stage_slice_T<TT>(Xt, Xg + (long long)n0 * D, min(N - n0, ldk), D, d, ldk, LDV, tid, WAVES * 64);
    __syncthreads();
    if (sc) {
      const int part = tid % 16;                         // 16 threads per feature row
      for (int t = tid / 16; t < 16 * TT; t += WAVES * 4) {
        float acc_s = 0.f;
        for (int n = part * 8; n < ldk; n += 128) {
          const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(Xt + t * LDV + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc_s += (float)v8[e];
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) acc_s += __shfl_xor(acc_s, m, 64);
        if (part == 0) colsum_s[t] += acc_s;
      }
      __syncthreads();
    }
  };
  const bf16_t* Mb = M + (long long)(bz / mdiv) * N * ld;      // (wide heads: mdiv slices of 96 features share the head's map)
  const bool vec = (d % 4 == 0) && (D % 4 == 0);
  const int lrow = lane >> 3, lch = (lane & 7) * 8;
  const int nstrips = (N + 63) >> 6;
  const int q = l15 >> 2, pq = l15 & 3;
  // one strip of 64 map columns against the staged rows: map rows rbase + [0, nrows), nrows a multiple of 32
  auto run = [&](int strip, int rbase, int nrows, f32x4 (&acc)[4][TT]) {
    const int jc = min(strip * 64 + lch, ld - 8);        // columns >= ld: any finite data, never stored
    // register ring two 32-row steps (8 x 16 B per lane) ahead; clamped unconditional loads: rows >= N meet zeros of Xt
    // (named registers, not an array: hipcc left an indexed ring in scratch memory)
    uint4 p0, p1, p2, p3, q0, q1, q2, q3;
    auto fetch = [&](int i0, uint4& m0, uint4& m1, uint4& m2, uint4& m3) {
      m0 = *reinterpret_cast<const uint4*>(Mb + (long long)min(rbase + i0 + lrow, N - 1) * ld + jc);
      m1 = *reinterpret_cast<const uint4*>(Mb + (long long)min(rbase + i0 + 8 + lrow, N - 1) * ld + jc);
      m2 = *reinterpret_cast<const uint4*>(Mb + (long long)min(rbase + i0 + 16 + lrow, N - 1) * ld + jc);
      m3 = *reinterpret_cast<const uint4*>(Mb + (long long)min(rbase + i0 + 24 + lrow, N - 1) * ld + jc);
    };
    auto put = [&](const uint4& m0, const uint4& m1, const uint4& m2, const uint4& m3) {
      *reinterpret_cast<uint4*>(T + lrow * MP_LDT + lch) = m0;
      *reinterpret_cast<uint4*>(T + (8 + lrow) * MP_LDT + lch) = m1;
      *reinterpret_cast<uint4*>(T + (16 + lrow) * MP_LDT + lch) = m2;
      *reinterpret_cast<uint4*>(T + (24 + lrow) * MP_LDT + lch) = m3;
    };
    auto mult = [&](int i0) {
      // A operand: Xt rows t, k-slots = staged rows i0 + 8 lg + e
      bf16x8 xa[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        if constexpr (NAT) xa[tt] = tr_pair<P>(Xt + (i0 + 8 * lg + q) * P + 16 * tt + 4 * pq);
        else xa[tt] = *reinterpret_cast<const bf16x8*>(Xt + (16 * tt + l15) * LDV + i0 + 8 * lg);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bf16_t* tb = T + (8 * lg + q) * MP_LDT + 16 * u + 4 * pq;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)tb);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tb + 4 * MP_LDT));
        const s16x8 t8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x8 bm = __builtin_bit_cast(bf16x8, t8);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[tt], bm, acc[u][tt], 0, 0, 0);
      }
    };
    auto step = [&](int i0, uint4& m0, uint4& m1, uint4& m2, uint4& m3) {
      put(m0, m1, m2, m3);
      fetch(i0 + 64, m0, m1, m2, m3);
      mult(i0);
    };
    fetch(0, p0, p1, p2, p3);
    fetch(32, q0, q1, q2, q3);
    // branch-free main loop (counted vmcnt waits), then the odd last step
    const int nfull = nrows / 64 * 64;
    for (int i00 = 0; i00 < nfull; i00 += 64) {
      step(i00, p0, p1, p2, p3);
      step(i00 + 32, q0, q1, q2, q3);
    }
    if (nfull < nrows) { put(p0, p1, p2, p3); mult(nfull); }
  };
  // C[row = t = 4 lg + r (+16)][col = token j0 + 16 u + l15]
  auto finish = [&](int strip, f32x4 (&acc)[4][TT]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = strip * 64 + 16 * u + l15;
      if (j < N) {
        bf16_t* orow = out + ((long long)b * N + j) * D + g * d;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          f32x4 o = acc[u][tt];
          if (sc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaf(a_sc, o[r], a_kp * colsum_s[16 * tt + 4 * lg + r]);
          }
          store_t4(orow, 16 * tt + 4 * lg, d, vec, o);
        }
      }
    }
  };
  if constexpr (!CHUNKED) {
    __syncthreads();
    stage(0);
    __syncthreads();
    for (int strip = blockIdx.x * WAVES + wave; strip < nstrips; strip += gridDim.x * WAVES) {
      f32x4 acc[4][TT];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[u][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      run(strip, 0, (N + 31) & ~31, acc);
      finish(strip, acc);
    }
  } else {
    const int strip = blockIdx.x * WAVES + wave;    // one strip per wave; every wave takes part in the staging
    f32x4 acc[4][TT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[u][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < N; n0 += mp_ch(TT)) {
      __syncthreads();
      stage(n0);
      __syncthreads();
      if (strip < nstrips) run(strip, n0, (min(N - n0, mp_ch(TT)) + 31) & ~31, acc);
    }
    if (strip < nstrips) finish(strip, acc);
  }
}

constexpr size_t mp_image_bytes(int tt, int ldk) {
  return mp_natural(tt) ? (size_t)ldk * mp_pitch(tt) * 2 : (size_t)16 * tt * (ldk + 8) * 2;
}
template <bool COLS, int WAVES, int TT, bool CHUNKED>
int launch_map_prod_w(const void* M, const void* X, void* out, const float* sc, const float* kappa, int B, int N, int D, int H, int ld,
                      int nsplit, int mdiv, hipStream_t st) {
  const int d = D / H;
  const int ldk = CHUNKED ? mp_ch(TT) : ((N + 63) & ~63);
  const size_t lds = mp_image_bytes(TT, ldk) + (size_t)WAVES * (COLS ? 32 : 16) * MP_LDT * 2;
  auto kern = COLS ? attn_map_cols_kernel<WAVES, TT, CHUNKED> : attn_map_rows_kernel<WAVES, TT, CHUNKED>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { vu_set_error("attn_map_prod: cannot reserve %zu bytes of LDS", lds); return VU_ELAUNCH; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nsplit, (unsigned)(B * H)), dim3(WAVES * 64), lds, st, (const bf16_t*)M,
                     (const bf16_t*)X, (bf16_t*)out, sc, kappa, N, D, H, d, ld, mdiv);
  if (vu_prof_on()) vu_prof_note(COLS ? "attn_map_cols_kernel" : "attn_map_rows_kernel", 2.0 * B * H * (double)N * N * d,
                                 ((double)B * H * N * ld + 2.0 * B * N * D) * 2.0);
  return vu_check_launch("vu_attn_map_prod");
}

template <bool COLS, int TT>
int launch_map_prod(const void* M, const void* X, void* out, const float* sc, const float* kappa, int B, int N, int D, int H, int ld,
                    int mdiv, hipStream_t st) {
  const int units = COLS ? (N + 63) / 64 : (N + 15) / 16;       // strips / row tiles per (sample, head)
  if (mp_image_bytes(TT, (N + 63) & ~63) + 8 * 32 * MP_LDT * 2 > 150 * 1024) {
    // the head slice does not fit one LDS image: chunked form, one unit per wave, 7 waves per workgroup
    return launch_map_prod_w<COLS, 7, TT, true>(M, X, out, sc, kappa, B, N, D, H, ld, (units + 6) / 7, mdiv, st);
  }
  // waves per workgroup: as many as there are units, in whole rounds where possible
  const int waves = units <= 4 ? 4 : ((units % 7 == 0 || (units > 8 && units <= 14)) ? 7 : 8);
  int nsplit = (512 + B * H - 1) / (B * H);
  const int maxsplit = (units + waves - 1) / waves;
  if (nsplit > maxsplit) nsplit = maxsplit;
  if (nsplit < 1) nsplit = 1;
  if (waves == 4) return launch_map_prod_w<COLS, 4, TT, false>(M, X, out, sc, kappa, B, N, D, H, ld, nsplit, mdiv, st);
  if (waves == 7) return launch_map_prod_w<COLS, 7, TT, false>(M, X, out, sc, kappa, B, N, D, H, ld, nsplit, mdiv, st);
  return launch_map_prod_w<COLS, 8, TT, false>(M, X, out, sc, kappa, B, N, D, H, ld, nsplit, mdiv, st);
}

}  // namespace

// returns VU_OK, a negative error, or 1 when the shape is not covered (the caller uses the batched GEMM)
int vu_k_attn_map_prod(int dtype, int cols, const void* M, const void* X, void* out, const float* sc, const float* kappa,
                       int B, int N, int D, int H, int ld, hipStream_t st) {
  const int d = D / H;
  if (dtype != 1 || N < 32 || ld % 8 != 0) return 1;
  // heads wider than 96 features (level 0: d = 384; Lite level 0: d = 192) run as d / 96 slices of 96 that share the head's
  // map: to the kernels a slice is a head of its own (H * mdiv heads of 96), only the map index is divided by mdiv
  int mdiv = 1, Hk = H;
  if (d > 96) {
    if (d % 96 != 0 || N > 256) return 1;
    mdiv = d / 96; Hk = H * mdiv;
  }
  const int dk = d / mdiv;
  const int tt = dk <= 32 ? 2 : 6;
  if (tt == 2) return cols ? launch_map_prod<true, 2>(M, X, out, sc, kappa, B, N, D, Hk, ld, mdiv, st) : launch_map_prod<false, 2>(M, X, out, sc, kappa, B, N, D, Hk, ld, mdiv, st);
  return cols ? launch_map_prod<true, 6>(M, X, out, sc, kappa, B, N, D, Hk, ld, mdiv, st) : launch_map_prod<false, 6>(M, X, out, sc, kappa, B, N, D, Hk, ld, mdiv, st);
}
