// Strided / batched MFMA GEMM for gfx950 with fused epilogues.
//
//   C[z](m,n) (+)= epilogue( alpha * sum_k A[z](m,k) * B[z](k,n) )
//
// Operands are addressed with element strides so the same kernel serves the projection /
// feed-forward linears, the per-head attention products (heads are column slices of the
// token-major (B,N,D) activations) and every backward product.  Each operand is either
// K-contiguous ("N" form: LDS image [row][k], fragments by ds_read_b128) or row-contiguous
// ("T" form: LDS image [k][row], fragments by ds_read_b64_tr_b16 - the gfx950 transposing LDS
// read - so no scattered LDS writes are needed).  T = __bf16 uses v_mfma_f32_16x16x32_bf16,
// T = float uses the exact-fp32 v_mfma_f32_16x16x4_f32 (parity mode).
//
// 256 threads = 4 waves in a 2x2 arrangement; wave tile (BM/2)x(BN/2) of 16x16 MFMA tiles;
// BK = 32..128 (template); register-prefetched global loads (tile t+1 in flight while tile t is multiplied).
#pragma once
#include "vu_common.h"

enum { VU_ACT_NONE = 0, VU_ACT_GELU = 1, VU_ACT_DGELU = 2 };

// Scratch for deterministic split-K (vu_tsgemm.hip): a caller that owns a workspace (the model executor) lends a slab for the
// duration of its call; launches made in between write their K-slice partials there with plain stores and add them up in a
// fixed order with one small kernel, instead of float atomics on the output.  Per-thread, scoped: set before, cleared after.
void vu_gemm_set_scratch(void* p, size_t bytes);
void vu_gemm_get_scratch(void** p, size_t* bytes);
// A second, larger region for the K-slice partials of the skinny weight gradients (vu_tsgemm.hip): products take pieces of it one
// after the other and QUEUE their fixed-order reductions; vu_tsgemm_flush launches all queued reductions as one kernel.  Whoever
// lends the arena flushes before anything outside its call reads the outputs, and before taking the arena back.
void vu_tsgemm_set_arena(void* p, size_t bytes);
int vu_tsgemm_flush(hipStream_t st);
// The same arena also takes the per-workgroup partial sums of OTHER weight-gradient kernels (the q / k / v convolutions' Gram and
// stencil forms, the head-mix gradients of the map-backward kernels): their fixed-order tails - one wave per output element over
// the workgroups' partials - were 21 launches of 5 - 6 us each per Base backward; queued here they are ONE launch at the flush.
// vu_deferred_take returns nullptr when no arena is lent (stand-alone op calls: the kernel's own reduce launch runs at once);
// the summation order of a queued reduction is the one of its stand-alone kernel (bit-identical results).
enum { VU_DEFRED_TZW = 0, VU_DEFRED_WGRAD_MM = 1, VU_DEFRED_WGRAD3 = 2, VU_DEFRED_MAP = 3 };
struct vu_defred {
  int kind, blk0;                // blk0: set by the queue
  int nblocks, n, hh, nconv;     // partial rows; MAP: outputs n = hh + h; WGRAD3: convolutions in the set
  const float* part;
  float* dst[6];                 // TZW / WGRAD_MM: dwq, dwk, dwv; WGRAD3: dw[3], dbias[3]; MAP: dW, dc
};
float* vu_deferred_take(size_t floats, hipStream_t st);
void vu_deferred_push(const vu_defred& d);

struct vu_gemm_args {
  const void* A; const void* B; void* C;
  int M, N, K;
  long long sAm, sAk, sBk, sBn, ldc;
  int Z1, Z2;  // batch count = Z1*Z2, z = z1*Z2 + z2
  long long sA1, sA2, sB1, sB2, sC1, sC2;
  float alpha;
  const float* bias;   // [N] fp32, or null
  int act;             // VU_ACT_*: GELU stores act(x) in C and x in aux; DGELU multiplies by gelu'(aux)
  void* aux;           // T, C layout
  const void* addend;  // T, C layout: C = result + addend
  int accumulate;      // C += result   (float C only)
  int dropout;         // apply dropout(rng) to result; element index = (z*M + m)*N + n
  vu_rng rng;
  int vecA, vecB;      // 16-byte vector loads legal for A / B
  int swap;            // operands were exchanged by the launcher: the kernel computes C^T (vector stores)
  int vecC;            // 4-element vector access to C rows is aligned
  int vec8;            // 8-element (16-byte) access to C / aux / addend rows is aligned and the row length is a multiple of 8
  float* colsum;       // optional: += column sums over k of an operand (bias gradients), see colsum_side
  int colsum_side;     // kernel space: 1 = sum_k A(m,k) -> colsum[m] ; 2 = sum_k B(k,n) -> colsum[n]
  int ksplit;          // >1: K is split over blockIdx.z and C is accumulated with float atomics
};

template <typename T> struct vu_vec { static constexpr int N = 16 / sizeof(T); };

template <typename T, typename TC> struct vu_epi_ctx {
  vu_rng rng; TC* Cb; T* auxb; const T* addb; bool lead; int z;
};

// Kernel space: rows km (the lane's 4 accumulator registers are 4 CONSECUTIVE rows), column kn.
// Plain form: C[km][kn], scalar stores.
template <typename T, typename TC>
__device__ __forceinline__ void vu_epilogue_plain(const vu_gemm_args& g, const vu_epi_ctx<T, TC>& ec, f32x4 a, int km0, int kn) {
  if (kn >= g.N || km0 >= g.M) return;
  const float bv = (g.bias && ec.lead) ? g.bias[kn] : 0.f;
  const float av[4] = {a[0], a[1], a[2], a[3]};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = km0 + r;
    if (m < g.M) {
      float v = av[r] * g.alpha + bv;
      const long long o = (long long)m * g.ldc + kn;
      if (g.act == VU_ACT_GELU) { vu_st(ec.auxb + o, v); v = vu_gelu(v); }
      else if (g.act == VU_ACT_DGELU) { v *= vu_gelu_grad(vu_ld(ec.auxb + o)); }
      if (g.dropout) {
        const uint64_t idx = ((uint64_t)ec.z * g.M + m) * (uint64_t)g.N + kn;
        v = vu_keep(ec.rng, idx) ? v * ec.rng.inv_keep : 0.f;
      }
      if (ec.addb) v += vu_ld(ec.addb + o);
      if constexpr (sizeof(TC) == 4) {
        float* cp = (float*)ec.Cb + o;
        if (g.ksplit > 1) atomicAdd(cp, v); else if (g.accumulate) *cp += v; else *cp = v;
      } else {
        vu_st((T*)ec.Cb + o, v);
      }
    }
  }
}

// Swapped form (the launcher exchanged the operands, the kernel computes C^T): kernel rows are the
// ORIGINAL columns, so the 4 registers are 4 consecutive elements of one row of C: vector access.
template <typename T, typename TC>
__device__ __forceinline__ void vu_epilogue_swapped(const vu_gemm_args& g, const vu_epi_ctx<T, TC>& ec, f32x4 a, int km0, int kn) {
  if (kn >= g.N || km0 >= g.M) return;
  const long long o = (long long)kn * g.ldc + km0;   // original element (row kn, columns km0..km0+3)
  const int nv = g.M - km0 < 4 ? g.M - km0 : 4;
  float v0 = a[0] * g.alpha, v1 = a[1] * g.alpha, v2 = a[2] * g.alpha, v3 = a[3] * g.alpha;
  if (g.bias && ec.lead) {
    v0 += g.bias[km0];
    if (nv > 1) v1 += g.bias[km0 + 1];
    if (nv > 2) v2 += g.bias[km0 + 2];
    if (nv > 3) v3 += g.bias[km0 + 3];
  }
  const bool vec = (nv == 4) && g.vecC;
  if (g.act == VU_ACT_GELU) {
    if (vec) { vu_f4 t4 = {{v0, v1, v2, v3}}; vu_st4(ec.auxb + o, t4); }
    else { vu_st(ec.auxb + o, v0); if (nv > 1) vu_st(ec.auxb + o + 1, v1); if (nv > 2) vu_st(ec.auxb + o + 2, v2); if (nv > 3) vu_st(ec.auxb + o + 3, v3); }
    v0 = vu_gelu(v0); v1 = vu_gelu(v1); v2 = vu_gelu(v2); v3 = vu_gelu(v3);
  } else if (g.act == VU_ACT_DGELU) {
    if (vec) { const vu_f4 t4 = vu_ld4(ec.auxb + o); v0 *= vu_gelu_grad(t4.v[0]); v1 *= vu_gelu_grad(t4.v[1]); v2 *= vu_gelu_grad(t4.v[2]); v3 *= vu_gelu_grad(t4.v[3]); }
    else {
      v0 *= vu_gelu_grad(vu_ld(ec.auxb + o));
      if (nv > 1) v1 *= vu_gelu_grad(vu_ld(ec.auxb + o + 1));
      if (nv > 2) v2 *= vu_gelu_grad(vu_ld(ec.auxb + o + 2));
      if (nv > 3) v3 *= vu_gelu_grad(vu_ld(ec.auxb + o + 3));
    }
  }
  if (g.dropout) {
    const uint64_t idx = ((uint64_t)ec.z * g.N + kn) * (uint64_t)g.M + km0;
    v0 = vu_keep(ec.rng, idx) ? v0 * ec.rng.inv_keep : 0.f;
    v1 = vu_keep(ec.rng, idx + 1) ? v1 * ec.rng.inv_keep : 0.f;
    v2 = vu_keep(ec.rng, idx + 2) ? v2 * ec.rng.inv_keep : 0.f;
    v3 = vu_keep(ec.rng, idx + 3) ? v3 * ec.rng.inv_keep : 0.f;
  }
  if (ec.addb) {
    if (vec) { const vu_f4 t4 = vu_ld4(ec.addb + o); v0 += t4.v[0]; v1 += t4.v[1]; v2 += t4.v[2]; v3 += t4.v[3]; }
    else {
      v0 += vu_ld(ec.addb + o);
      if (nv > 1) v1 += vu_ld(ec.addb + o + 1);
      if (nv > 2) v2 += vu_ld(ec.addb + o + 2);
      if (nv > 3) v3 += vu_ld(ec.addb + o + 3);
    }
  }
  if constexpr (sizeof(TC) == 4) {
    float* cp = (float*)ec.Cb + o;
    if (g.ksplit > 1) {
      atomicAdd(cp, v0);
      if (nv > 1) atomicAdd(cp + 1, v1);
      if (nv > 2) atomicAdd(cp + 2, v2);
      if (nv > 3) atomicAdd(cp + 3, v3);
    } else if (vec) {
      float4 c4 = make_float4(v0, v1, v2, v3);
      if (g.accumulate) { const float4 p4 = *reinterpret_cast<const float4*>(cp); c4.x += p4.x; c4.y += p4.y; c4.z += p4.z; c4.w += p4.w; }
      *reinterpret_cast<float4*>(cp) = c4;
    } else {
      if (g.accumulate) { v0 += cp[0]; if (nv > 1) v1 += cp[1]; if (nv > 2) v2 += cp[2]; if (nv > 3) v3 += cp[3]; }
      cp[0] = v0; if (nv > 1) cp[1] = v1; if (nv > 2) cp[2] = v2; if (nv > 3) cp[3] = v3;
    }
  } else {
    T* cp = (T*)ec.Cb + o;
    if (vec) { vu_f4 t4 = {{v0, v1, v2, v3}}; vu_st4(cp, t4); }
    else { vu_st(cp, v0); if (nv > 1) vu_st(cp + 1, v1); if (nv > 2) vu_st(cp + 2, v2); if (nv > 3) vu_st(cp + 3, v3); }
  }
}

// PD: k-tiles of global loads kept in flight per thread (register ring).  1 is the classic one-tile prefetch; the small
// tiles used for long-K products with few output tiles (64 x 64: 98 - 196 workgroups on 256 CUs) take 4, because a stream
// that few workgroups read is bound by bytes in flight (one 16 KB k-tile per ~2 us of loaded-chip latency = 8 GB/s per CU).
template <typename T, typename TC, bool TA, bool TB, int BM, int BN, int BK, int PD = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? 3 : 1, 8))) void vu_gemm_kernel(const vu_gemm_args g) {
  constexpr int VEC = vu_vec<T>::N;            // elements per 16 B
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int PADK = IS_BF16 ? 8 : 4;        // [row][k] images: row stride 80 B / 144 B
  constexpr int PADR = IS_BF16 ? 16 : 4;       // [k][row] images
  constexpr int LDA = TA ? (BM + PADR) : (BK + PADK);
  constexpr int LDB = TB ? (BN + PADR) : (BK + PADK);
  constexpr int A_ELEMS = TA ? BK * LDA : BM * LDA;
  constexpr int B_ELEMS = TB ? BK * LDB : BN * LDB;
  constexpr int TM = BM / 32, TN = BN / 32;    // 16x16 tiles per wave
  constexpr int NCA = BM * BK / VEC / 256;     // 16-B chunks per thread per k-tile
  constexpr int NCB = (BN * BK / VEC + 255) / 256;
  static_assert(NCA >= 1, "tile too small");

  __shared__ __attribute__((aligned(16))) T smem[A_ELEMS + B_ELEMS];
  T* As = smem;
  T* Bs = smem + A_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  // (an XCD-aware remap of the tile order was measured: no gain on the 3072-class GEMMs, tools/gemm_bench.py)
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int z = blockIdx.y, z1 = z / g.Z2, z2 = z % g.Z2;
  const int m_base = tile_m * BM, n_base = tile_n * BN;

  const T* Ab = (const T*)g.A + z1 * g.sA1 + z2 * g.sA2;
  const T* Bb = (const T*)g.B + z1 * g.sB1 + z2 * g.sB2;

  // ---- global -> register staging ---------------------------------------------------------
  uint4 ring_a[PD][NCA], ring_b[PD][NCB];

  auto load_tile = [&](int k0, uint4 (&ra)[NCA], uint4 (&rb)[NCB]) {
#pragma unroll
    for (int c = 0; c < NCA; ++c) {
      const int ch = tid + c * 256;
      int row, kk;  // row = m index in tile, kk = k index in tile of the chunk's first element
      if (TA) { kk = ch / (BM / VEC); row = (ch % (BM / VEC)) * VEC; }
      else    { row = ch / (BK / VEC); kk = (ch % (BK / VEC)) * VEC; }
      const int m = m_base + row, k = k0 + kk;
      uint4 v = make_uint4(0, 0, 0, 0);
      const bool full = TA ? (k < g.K && m + VEC <= g.M) : (m < g.M && k + VEC <= g.K);
      if (full && g.vecA) {
        v = *reinterpret_cast<const uint4*>(Ab + (long long)m * g.sAm + (long long)k * g.sAk);
      } else {
        alignas(16) T tmp[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int mm = TA ? m + e : m, kx = TA ? k : k + e;
          tmp[e] = (mm < g.M && kx < g.K) ? Ab[(long long)mm * g.sAm + (long long)kx * g.sAk] : (T)0.f;
        }
        v = *reinterpret_cast<uint4*>(tmp);
      }
      ra[c] = v;
    }
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const int ch = tid + c * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ch < BN * BK / VEC) {
        int col, kk;
        if (TB) { kk = ch / (BN / VEC); col = (ch % (BN / VEC)) * VEC; }
        else    { col = ch / (BK / VEC); kk = (ch % (BK / VEC)) * VEC; }
        const int n = n_base + col, k = k0 + kk;
        const bool full = TB ? (k < g.K && n + VEC <= g.N) : (n < g.N && k + VEC <= g.K);
        if (full && g.vecB) {
          v = *reinterpret_cast<const uint4*>(Bb + (long long)k * g.sBk + (long long)n * g.sBn);
        } else {
          alignas(16) T tmp[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const int nn = TB ? n + e : n, kx = TB ? k : k + e;
            tmp[e] = (nn < g.N && kx < g.K) ? Bb[(long long)kx * g.sBk + (long long)nn * g.sBn] : (T)0.f;
          }
          v = *reinterpret_cast<uint4*>(tmp);
        }
      }
      rb[c] = v;
    }
  };
  auto store_tile = [&](const uint4 (&ra)[NCA], const uint4 (&rb)[NCB]) {
#pragma unroll
    for (int c = 0; c < NCA; ++c) {
      const int ch = tid + c * 256;
      int row, kk;
      if (TA) { kk = ch / (BM / VEC); row = (ch % (BM / VEC)) * VEC; *reinterpret_cast<uint4*>(&As[kk * LDA + row]) = ra[c]; }
      else    { row = ch / (BK / VEC); kk = (ch % (BK / VEC)) * VEC; *reinterpret_cast<uint4*>(&As[row * LDA + kk]) = ra[c]; }
    }
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const int ch = tid + c * 256;
      if (ch < BN * BK / VEC) {
        int col, kk;
        if (TB) { kk = ch / (BN / VEC); col = (ch % (BN / VEC)) * VEC; *reinterpret_cast<uint4*>(&Bs[kk * LDB + col]) = rb[c]; }
        else    { col = ch / (BK / VEC); kk = (ch % (BK / VEC)) * VEC; *reinterpret_cast<uint4*>(&Bs[col * LDB + kk]) = rb[c]; }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lg = lane >> 4;
  const int nk_all = (g.K + BK - 1) / BK;
  int kt0 = 0, nk = nk_all;
  if (g.ksplit > 1) {   // split-K: this block owns k-tiles [kt0, nk)
    const int per = (nk_all + g.ksplit - 1) / g.ksplit;
    kt0 = blockIdx.z * per;
    nk = kt0 + per < nk_all ? kt0 + per : nk_all;
  }
  float csum = 0.f;   // bias-gradient column sum carried by the first tile column / row of blocks
  const bool do_cs = g.colsum && ((g.colsum_side == 1 && tile_n == 0 && tid < BM) || (g.colsum_side == 2 && tile_m == 0 && tid < BN));
#pragma unroll
  for (int s_ = 0; s_ < PD; ++s_)
    if (kt0 + s_ < nk) load_tile((kt0 + s_) * BK, ring_a[s_], ring_b[s_]);
  for (int ktb = kt0; ktb < nk; ktb += PD) {
#pragma unroll
   for (int s_ = 0; s_ < PD; ++s_) {
    const int kt = ktb + s_;
    if (kt >= nk) break;                 // workgroup-uniform
    store_tile(ring_a[s_], ring_b[s_]);
    __syncthreads();
    if (kt + PD < nk) load_tile((kt + PD) * BK, ring_a[s_], ring_b[s_]);
    if (do_cs) {
      if (g.colsum_side == 1) {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) csum += (float)(TA ? As[kk * LDA + tid] : As[tid * LDA + kk]);
      } else {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) csum += (float)(TB ? Bs[kk * LDB + tid] : Bs[tid * LDB + kk]);
      }
    }
    if constexpr (IS_BF16) {
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
#pragma unroll
      for (int k32 = 0; k32 < BK; k32 += 32) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r0 = wm * (BM / 2) + i * 16;
        if constexpr (TA) {
          const int q = l15 >> 2, p = l15 & 3;
          const bf16_t* a0 = (const bf16_t*)&As[(k32 + 8 * lg + q) * LDA + r0 + 4 * p];
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a0);
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 4 * LDA));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[i] = __builtin_bit_cast(bf16x8, t);
        } else {
          af[i] = *reinterpret_cast<const bf16x8*>(&As[(r0 + l15) * LDA + k32 + 8 * lg]);
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c0 = wn * (BN / 2) + j * 16;
        if constexpr (TB) {
          const int q = l15 >> 2, p = l15 & 3;
          const bf16_t* b0 = (const bf16_t*)&Bs[(k32 + 8 * lg + q) * LDB + c0 + 4 * p];
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)b0);
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 4 * LDB));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfr[j] = __builtin_bit_cast(bf16x8, t);
        } else {
          bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(c0 + l15) * LDB + k32 + 8 * lg]);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float af[TM], bfr[TN];
        const int kq = ks * 4 + lg;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = wm * (BM / 2) + i * 16 + l15;
          af[i] = TA ? (float)As[kq * LDA + r] : (float)As[r * LDA + kq];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int c = wn * (BN / 2) + j * 16 + l15;
          bfr[j] = TB ? (float)Bs[kq * LDB + c] : (float)Bs[c * LDB + kq];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
   }
  }

  if (do_cs) {
    const int idx = (g.colsum_side == 1 ? m_base : n_base) + tid;
    if (idx < (g.colsum_side == 1 ? g.M : g.N)) {
      if (g.ksplit > 1) atomicAdd(g.colsum + idx, csum); else g.colsum[idx] += csum;
    }
  }
  // ---- epilogue -----------------------------------------------------------------------------
  if constexpr (IS_BF16 && sizeof(TC) == 2 && BM == 128 && BN == 128) {
    if (g.vec8) {
      // bf16 output of the big tile: the accumulators go through LDS (fp32, two passes of 64 ORIGINAL rows) so that
      // every global access of the epilogue - C, the GELU pre-activation, the residual addend - is a 16-byte
      // access inside a full 256-byte row segment.  (The direct form writes 16 rows x 32 B per instruction; on a
      // K = 64 GEMM that epilogue was 20 of 23 us.)
      constexpr int LDC = 132;
      float* Ct = reinterpret_cast<float*>(smem);
      static_assert(64 * LDC * 4 <= (A_ELEMS + B_ELEMS) * (int)sizeof(T), "C staging tile must fit the operand buffers");
      const vu_rng rng = g.dropout ? vu_rng_resolve(g.rng) : g.rng;
      const long long coff = z1 * g.sC1 + z2 * g.sC2;
      T* Cb = (T*)g.C + coff;
      T* auxb = g.aux ? (T*)g.aux + coff : nullptr;
      const T* addb = g.addend ? (const T*)g.addend + coff : nullptr;
      const int Rtot = g.swap ? g.N : g.M, Ctot = g.swap ? g.M : g.N;          // original rows / columns
      const int row0 = g.swap ? n_base : m_base, col0 = g.swap ? m_base : n_base;
      const bool lead = blockIdx.z == 0;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (p) __syncthreads();
        if (g.swap) {
          if (wn == p) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const float4 v4 = make_float4(acc[i][j][0] * g.alpha, acc[i][j][1] * g.alpha, acc[i][j][2] * g.alpha, acc[i][j][3] * g.alpha);
                *reinterpret_cast<float4*>(&Ct[(j * 16 + l15) * LDC + wm * 64 + i * 16 + lg * 4]) = v4;
              }
          }
        } else {
          if (wm == p) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ct[(i * 16 + lg * 4 + r) * LDC + wn * 64 + j * 16 + l15] = acc[i][j][r] * g.alpha;
          }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int rl = it * 16 + (tid >> 4), c8 = (tid & 15) * 8;
          const int orow = row0 + 64 * p + rl, ocol = col0 + c8;
          if (orow < Rtot && ocol < Ctot) {
            float v[8];
            const float4 a0 = *reinterpret_cast<const float4*>(&Ct[rl * LDC + c8]), a1 = *reinterpret_cast<const float4*>(&Ct[rl * LDC + c8 + 4]);
            v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
            if (g.bias && lead) {
              const float4 b0 = *reinterpret_cast<const float4*>(g.bias + ocol), b1 = *reinterpret_cast<const float4*>(g.bias + ocol + 4);
              v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            const long long o = (long long)orow * g.ldc + ocol;
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
              const uint64_t idx = ((uint64_t)z * Rtot + orow) * (uint64_t)Ctot + ocol;
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
      return;
    }
  }
  vu_epi_ctx<T, TC> ec;
  ec.rng = g.dropout ? vu_rng_resolve(g.rng) : g.rng;
  const long long coff = z1 * g.sC1 + z2 * g.sC2;
  ec.Cb = (TC*)g.C + coff;
  ec.auxb = g.aux ? (T*)g.aux + coff : nullptr;
  ec.addb = g.addend ? (const T*)g.addend + coff : nullptr;
  ec.lead = blockIdx.z == 0;
  ec.z = z;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int kn = n_base + wn * (BN / 2) + j * 16 + l15;
      const int km0 = m_base + wm * (BM / 2) + i * 16 + lg * 4;
      if (g.swap) vu_epilogue_swapped<T, TC>(g, ec, acc[i][j], km0, kn);
      else vu_epilogue_plain<T, TC>(g, ec, acc[i][j], km0, kn);
    }
}
