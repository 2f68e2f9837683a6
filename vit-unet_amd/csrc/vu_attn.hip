// Fused re-attention kernels for gfx950.
//
// attn_scores_kernel (K6+K7+K8, model.py:155-157): one workgroup = 64 query rows of one (batch,
// head); the head's whole K (N x d, always 2*P/h bytes = 37.6 KB in bf16 for 224x224x3) is staged
// in LDS once, each wave computes its 16 x N logits tile with MFMA into registers (N <= 784:
// 49 accumulator tiles), does the row softmax with 16-lane shuffles, draws the dropout mask from
// the counter hash and writes the sign-tagged probabilities - the (B,h,N,N) logits never touch
// HBM.  Algorithmic traffic: one write of the map (E*|T|) + q, k reads.
#include <type_traits>
#include "vu_kernels.h"

namespace {

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
__global__ __launch_bounds__(WAVES * 64) void attn_scores_kernel(const T* __restrict__ q, const T* __restrict__ k,
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
    const int total = N * chunks_per_row;
    for (int c = tid; c < total; c += WAVES * 64) {
      const int row = c / chunks_per_row, kc = (c % chunks_per_row) * VE;
      alignas(16) T tmp[VE];
      if (vec && kc + VE <= d) {
        *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(kb + (long long)row * D + kc);
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) tmp[e] = (kc + e < d) ? kb[(long long)row * D + kc + e] : (T)0.f;
      }
      *reinterpret_cast<uint4*>(&Ks[row * LDK + kc]) = *reinterpret_cast<uint4*>(tmp);
    }
  }
  // ---- this wave's q fragments (16 rows x DP) ------------------------------------------------
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
  __syncthreads();

  // Swapped product S^T = K Q^T: the accumulator of tile nt holds, for query i0+l15 (the lane's
  // column), keys nt*16 + lg*4 + r (r = 0..3) - four consecutive keys per lane, so the row
  // softmax reduces in-lane plus two shuffles, a dropout hash word serves an in-lane key pair,
  // and the store is one 4-element vector per tile.
  const int ntiles = (N + 15) >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nt < ntiles) {
      const int key = nt * 16 + l15;
      const bool kv = key < N;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        Frag kf;
        if constexpr (MM::FE == 1) kf = kv ? (float)Ks[key * LDK + ks * MM::KS + lg] : 0.f;
        else kf = kv ? *reinterpret_cast<const Frag*>(&Ks[key * LDK + ks * MM::KS + lg * MM::FE]) : MM::zero();
        acc[nt] = MM::mma(kf, qf[ks], acc[nt]);
      }
    }
  }
  if constexpr (!SOFTMAX) {   // plain product (dAhat = dO v^T in the backward): scaled vector stores
    const int i = i0 + l15;
    if (i < N) {
      T* prow = Ps + ((long long)bz * N + i) * ld;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt < ntiles) {
          const int j0 = nt * 16 + lg * 4;
          if (EXACT || j0 < ld) {
            vu_f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o.v[r] = (EXACT || j0 + r < N) ? acc[nt][r] * scale : 0.f;
            vu_st4(prow + j0, o);
          }
        }
      }
    }
    return;
  }
  // ---- row softmax (logits rounded to the storage type first, like the unfused path) ------------
  // Only the last key tile can be partial: full tiles take a mask-free path (wave-uniform branch).
  float mx = -INFINITY;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (nt < ntiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sv = acc[nt][r] * scale;
        if constexpr (sizeof(T) == 2) sv = (float)(bf16_t)sv;
        if constexpr (!EXACT) sv = (nt * 16 + lg * 4 + r < N) ? sv : -INFINITY;
        acc[nt][r] = sv;
        mx = fmaxf(mx, sv);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
  const float mxl = mx * 1.44269504088896340736f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (nt < ntiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = exp2f(acc[nt][r] * 1.44269504088896340736f - mxl);   // exp(-inf) = 0 for masked keys
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
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt < ntiles) {
        const int j0 = nt * 16 + lg * 4;
        if (EXACT || j0 < ld) {
          float o0 = acc[nt][0] * inv, o1 = acc[nt][1] * inv, o2 = acc[nt][2] * inv, o3 = acc[nt][3] * inv;
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
          vu_f4 o = {{o0, o1, o2, o3}};
          vu_st4(prow + j0, o);
        }
      }
    }
  }
}

template <typename T, int NT, int DP>
int launch_scores(const T* q, const T* k, T* Ps, int B, int N, int D, int H, int ld, float scale, vu_rng rng,
                  bool softmax, hipStream_t st) {
  const int d = D / H;
  constexpr int LDK = DP + (sizeof(T) == 2 ? 8 : 4);
  const size_t lds = (size_t)N * LDK * sizeof(T);
  constexpr int WAVES = NT > 13 ? 8 : 4;
  auto kern = softmax ? ((N % 16 == 0) ? attn_scores_kernel<T, NT, DP, WAVES, true, true> : attn_scores_kernel<T, NT, DP, WAVES, false, true>)
                      : ((N % 16 == 0) ? attn_scores_kernel<T, NT, DP, WAVES, true, false> : attn_scores_kernel<T, NT, DP, WAVES, false, false>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { vu_set_error("attn_scores: cannot reserve %zu bytes of LDS", lds); return VU_ELAUNCH; }
  }
  dim3 grid((unsigned)((N + WAVES * 16 - 1) / (WAVES * 16)), (unsigned)(B * H));
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, st, q, k, Ps, N, D, H, d, ld, scale, rng);
  if (vu_prof_on()) vu_prof_note(softmax ? "attn_scores_kernel" : "attn_dscores_kernel", 2.0 * B * H * (double)N * N * d,
                                 ((double)B * H * N * N + 2.0 * B * N * D) * sizeof(T));
  return vu_check_launch("vu_attn_scores");
}

template <typename T>
int dispatch_scores(const void* q, const void* k, void* Ps, int B, int N, int D, int H, int ld, float scale, vu_rng rng,
                    bool softmax, hipStream_t st) {
  const int d = D / H;
  const int dp = (d + 31) / 32 * 32;
  const int nt = (N + 15) / 16;
#define VU_SC(NTv, DPv) return launch_scores<T, NTv, DPv>((const T*)q, (const T*)k, (T*)Ps, B, N, D, H, ld, scale, rng, softmax, st)
  if (nt <= 4) {
    if (dp == 32) VU_SC(4, 32); if (dp == 64) VU_SC(4, 64); if (dp == 96) VU_SC(4, 96); if (dp == 128) VU_SC(4, 128);
    if (dp == 192) VU_SC(4, 192); if (dp == 384) VU_SC(4, 384);
  } else if (nt <= 13) {
    if (dp == 32) VU_SC(13, 32); if (dp == 64) VU_SC(13, 64); if (dp == 96) VU_SC(13, 96); if (dp == 128) VU_SC(13, 128);
  } else if (nt <= 49) {
    if (dp == 32) VU_SC(49, 32); if (dp == 64) VU_SC(49, 64);
  }
#undef VU_SC
  return 1;   // shape not covered: the caller falls back to GEMM + softmax kernels
}

}  // namespace

// returns VU_OK, a negative error, or 1 when the shape is not covered by the fused kernel
int vu_k_attn_scores(int dtype, const void* q, const void* k, void* Ps, int B, int N, int D, int H, int ld, float scale,
                     vu_rng rng, hipStream_t st) {
  const int d = D / H;
  const int dp = (d + 31) / 32 * 32;
  const size_t es = dtype == 0 ? 4 : 2;
  const size_t lds = (size_t)N * (dp + (es == 2 ? 8 : 4)) * es;
  if (lds > 150 * 1024) return 1;
  if ((double)B * H * N * (double)ld >= 4294967295.0) return 1;   // 32-bit mask index in the fused kernel
  if (dtype == 0) return dispatch_scores<float>(q, k, Ps, B, N, D, H, ld, scale, rng, true, st);
  return dispatch_scores<bf16_t>(q, k, Ps, B, N, D, H, ld, scale, rng, true, st);
}

// out[b,g,i,j] = scale * sum_t a[b,i,g*d+t] * bmat[b,j,g*d+t]   (dAhat = dO v^T); 1 = shape not covered
int vu_k_attn_outer(int dtype, const void* a, const void* bmat, void* out, int B, int N, int D, int H, int ld, float scale,
                    hipStream_t st) {
  const int d = D / H;
  const int dp = (d + 31) / 32 * 32;
  const size_t es = dtype == 0 ? 4 : 2;
  const size_t lds = (size_t)N * (dp + (es == 2 ? 8 : 4)) * es;
  if (lds > 150 * 1024) return 1;
  vu_rng none = vu_make_rng(0, 0, 0.f);
  if (dtype == 0) return dispatch_scores<float>(a, bmat, out, B, N, D, H, ld, scale, none, false, st);
  return dispatch_scores<bf16_t>(a, bmat, out, B, N, D, H, ld, scale, none, false, st);
}

// =============================================================================================
// map_bwd_row_kernel: backward of BatchNorm -> head mix -> dropout -> softmax on one map row for
// ALL heads, with the row held in registers (one read of the tagged probabilities and of dAhat,
// one write of dS over dAhat).  TPR threads per row, each owning 4 consecutive columns; a block
// of 256 threads handles 256/TPR rows per iteration and walks the rows persistently so that the
// head-mix weight gradient (h x h) stays in registers until one final reduction.
// =============================================================================================
namespace {

template <typename T, int H, int TPR>
__global__ __launch_bounds__(256) void map_bwd_row_kernel(const T* __restrict__ Ps, T* dA, const float* __restrict__ W,
                                                          const float* __restrict__ c, const float* __restrict__ gamma,
                                                          const float* __restrict__ stats, float* dW, float* dc,
                                                          long long rows, int N, int ld, float inv_keep, float scale) {
  constexpr int RPB = 256 / TPR;
  __shared__ float sW[H * H];
  __shared__ float sX[H * H + H];
  __shared__ float sG[3 * H];
  __shared__ float redd[4][H];
  __shared__ float red[4][H * H + H];
  for (int i = threadIdx.x; i < H * H; i += blockDim.x) { sW[i] = W[i]; sX[i] = W[i] * stats[H * H + 2 * H + i / H]; }
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    sX[H * H + i] = (c[i] - stats[H * H + H + i]) * stats[H * H + 2 * H + i];
    sG[i] = gamma[i] * stats[H * H + 2 * H + i];
    sG[H + i] = stats[H * H + 3 * H + i];
    sG[2 * H + i] = stats[H * H + 4 * H + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rsub = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int jc = t * 4;
  const long long hs = (long long)N * ld;
  float aW[H * H], ac[H];
#pragma unroll
  for (int i = 0; i < H * H; ++i) aW[i] = 0.f;
#pragma unroll
  for (int i = 0; i < H; ++i) ac[i] = 0.f;
  const long long nrow_iters = (rows + RPB - 1) / RPB;
  for (long long it = blockIdx.x; it < nrow_iters; it += gridDim.x) {
    const long long row = it * RPB + rsub;
    const bool live = row < rows && jc < ld;
    const long long b = live ? row / N : 0;
    const int i = live ? (int)(row - b * N) : 0;
    const long long off = (b * H * N + i) * (long long)ld + jc;
    float pv[H][4], dP[H][4];
    float delta[H];
#pragma unroll
    for (int h = 0; h < H; ++h) delta[h] = 0.f;
    if (live) {
      float dAh[H][4];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const vu_f4 v = vu_ld4(Ps + off + h * hs);
        const vu_f4 d = vu_ld4(dA + off + h * hs);
#pragma unroll
        for (int e = 0; e < 4; ++e) { pv[h][e] = v.v[e]; dAh[h][e] = d.v[e]; }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool cv = jc + e < N;
        float pt[H], dAg[H];
#pragma unroll
        for (int h = 0; h < H; ++h) pt[h] = pv[h][e] > 0.f ? pv[h][e] * inv_keep : 0.f;
#pragma unroll
        for (int g = 0; g < H; ++g) {
          float xh = sX[H * H + g];
#pragma unroll
          for (int h = 0; h < H; ++h) xh += sX[g * H + h] * pt[h];
          dAg[g] = cv ? sG[g] * (dAh[g][e] - sG[H + g] - xh * sG[2 * H + g]) : 0.f;
          ac[g] += dAg[g];
#pragma unroll
          for (int h = 0; h < H; ++h) aW[g * H + h] += dAg[g] * pt[h];
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float dp = 0.f;
#pragma unroll
          for (int g = 0; g < H; ++g) dp += sW[g * H + h] * dAg[g];
          dp = pv[h][e] > 0.f ? dp * inv_keep : 0.f;
          dP[h][e] = dp;
          delta[h] += dp * fabsf(pv[h][e]);
        }
      }
    }
    // delta[h] = sum over the row
    if constexpr (TPR == 64) {
#pragma unroll
      for (int h = 0; h < H; ++h) delta[h] = vu_wave_sum(delta[h]);
    } else {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float v = vu_wave_sum(delta[h]);
        if (lane == 0) redd[wave][h] = v;
      }
      __syncthreads();
#pragma unroll
      for (int h = 0; h < H; ++h) delta[h] = redd[0][h] + redd[1][h] + redd[2][h] + redd[3][h];
      __syncthreads();
    }
    if (live) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        vu_f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (jc + e < N) ? fabsf(pv[h][e]) * (dP[h][e] - delta[h]) * scale : 0.f;
        vu_st4(dA + off + h * hs, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < H * H; ++i) { const float v = vu_wave_sum(aW[i]); if (lane == 0) red[wave][i] = v; }
#pragma unroll
  for (int i = 0; i < H; ++i) { const float v = vu_wave_sum(ac[i]); if (lane == 0) red[wave][H * H + i] = v; }
  __syncthreads();
  for (int i = threadIdx.x; i < H * H + H; i += blockDim.x) {
    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i < H * H) atomicAdd(dW + i, v); else atomicAdd(dc + (i - H * H), v);
  }
}

// ---------------------------------------------------------------------------------------------
// map_bwd_split_kernel: same arithmetic, but the H heads of a position quad are SPLIT over LS
// adjacent lanes (HL = H/LS heads each).  Each lane keeps only its heads' data and its rows of
// the mix matrices in registers (<= 128 VGPRs -> 4 waves per SIMD instead of 1) and the lanes
// of a quad exchange the post-dropout probabilities and the BatchNorm-backward gradients with
// quad-permute DPP moves.  One 1024-thread block walks rows persistently; TPR threads per row.
// ---------------------------------------------------------------------------------------------
// value of lane (lane ^ K) inside the lane's quad, K = 1..3, as one DPP quad_perm move
template <int K>
__device__ __forceinline__ float quad_xor(float v) {
  constexpr int ctrl = K == 1 ? 0xB1 : (K == 2 ? 0x4E : 0x1B);   // quad_perm [1,0,3,2] / [2,3,0,1] / [3,2,1,0]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true));
}
template <int LS>
__device__ __forceinline__ float quad_get(float v, int k) {   // k is a compile-time constant after unrolling
  if (k == 1) return quad_xor<1>(v);
  if (k == 2) return quad_xor<2>(v);
  if (k == 3) return quad_xor<3>(v);
  return v;
}

template <typename T, int H, int LS, int TPR>
__global__ __launch_bounds__(1024) void map_bwd_split_kernel(const T* __restrict__ Ps, T* dA, const float* __restrict__ W,
                                                             const float* __restrict__ c, const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, float* dW, float* dc,
                                                             long long rows, int N, int ld, float inv_keep, float scale) {
  constexpr int HL = H / LS;
  constexpr int RPB = 1024 / TPR;          // rows per block iteration
  constexpr int WPR = TPR / 64;            // waves per row
  __shared__ float redd[16][H];            // per-wave partial deltas
  __shared__ float red[16][H * H + H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = threadIdx.x % LS;          // head group of this lane
  const int rsub = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int jc = (t / LS) * 4;             // first column of this lane's quad
  const long long hs = (long long)N * ld;
  // mix-matrix rows / columns of every head group, in that group's relative slot order, in LDS
  // (re-read per use: keeping them in registers pushes the 128-VGPR budget into scratch)
  __shared__ __attribute__((aligned(16))) float sXr[LS][HL][H];   // W[g][hq] * rstd_g
  __shared__ __attribute__((aligned(16))) float sWc[LS][H][HL];   // W[hq][g]
  __shared__ float sK[LS][HL][4];                                  // Xc, Gs, M1, M2
  for (int q0 = threadIdx.x; q0 < LS * HL * H; q0 += blockDim.x) {
    const int uu = q0 / (HL * H), j = (q0 / H) % HL, q = q0 % H;
    const int g = uu * HL + j, hq = (uu ^ (q / HL)) * HL + q % HL;
    sXr[uu][j][q] = W[g * H + hq] * stats[H * H + 2 * H + g];
    sWc[uu][q][j] = W[hq * H + g];
  }
  for (int q0 = threadIdx.x; q0 < LS * HL; q0 += blockDim.x) {
    const int g = q0;
    const float rstd = stats[H * H + 2 * H + g];
    sK[g / HL][g % HL][0] = (c[g] - stats[H * H + H + g]) * rstd;
    sK[g / HL][g % HL][1] = gamma[g] * rstd;
    sK[g / HL][g % HL][2] = stats[H * H + 3 * H + g];
    sK[g / HL][g % HL][3] = stats[H * H + 4 * H + g];
  }
  __syncthreads();
  float aW[HL][H], ac[HL];
#pragma unroll
  for (int j = 0; j < HL; ++j) { ac[j] = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) aW[j][h] = 0.f; }

  const long long nit = (rows + RPB - 1) / RPB;
  for (long long it = blockIdx.x; it < nit; it += gridDim.x) {
    const long long row = it * RPB + rsub;
    const bool live = row < rows && jc < ld;
    const long long b = live ? row / N : 0;
    const int i = live ? (int)(row - b * N) : 0;
    const long long off = ((b * H + u * HL) * N + i) * (long long)ld + jc;   // head u*HL of this lane
    float pv[HL][4], dAh[HL][4], dP[HL][4], delta[HL];
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      delta[j] = 0.f;
      vu_f4 v = {{0.f, 0.f, 0.f, 0.f}}, d = {{0.f, 0.f, 0.f, 0.f}};
      if (live) { v = vu_ld4(Ps + off + j * hs); d = vu_ld4(dA + off + j * hs); }
#pragma unroll
      for (int e = 0; e < 4; ++e) { pv[j][e] = v.v[e]; dAh[j][e] = d.v[e]; }
    }
    auto step = [&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      const bool cv = live && (jc + e < N);
      // post-dropout probabilities of ALL heads, in "relative" order: slot k*HL+j holds head
      // (u^k)*HL+j, i.e. slot 0.. are the lane's own heads, then the quad partners'
      float ptr_[H];
#pragma unroll
      for (int j = 0; j < HL; ++j) {
        const float own = pv[j][e] > 0.f ? pv[j][e] * inv_keep : 0.f;
        ptr_[j] = own;
#pragma unroll
        for (int k = 1; k < LS; ++k) ptr_[k * HL + j] = quad_get<LS>(own, k);
      }
      float dAg[HL];
      int zz = 0;
      asm volatile("" : "+v"(zz));          // opaque zero: keeps the LDS table reads inside the loop
      const int uz = u + zz;
#pragma unroll
      for (int j = 0; j < HL; ++j) {
        float xh = sK[uz][j][0];
#pragma unroll
        for (int q = 0; q < H; ++q) xh += sXr[uz][j][q] * ptr_[q];
        dAg[j] = cv ? sK[uz][j][1] * (dAh[j][e] - sK[uz][j][2] - xh * sK[uz][j][3]) : 0.f;
        ac[j] += dAg[j];
#pragma unroll
        for (int q = 0; q < H; ++q) aW[j][q] += dAg[j] * ptr_[q];
      }
      // dP~ of the lane's own heads needs dA of all heads (same relative order)
      float dpa[HL];
#pragma unroll
      for (int j = 0; j < HL; ++j) dpa[j] = 0.f;
#pragma unroll
      for (int jj = 0; jj < HL; ++jj) {
#pragma unroll
        for (int k = 0; k < LS; ++k) {
          const float o = (k == 0) ? dAg[jj] : quad_get<LS>(dAg[jj], k);
#pragma unroll
          for (int j = 0; j < HL; ++j) dpa[j] += sWc[uz][k * HL + jj][j] * o;
        }
      }
#pragma unroll
      for (int j = 0; j < HL; ++j) {
        const float dp = pv[j][e] > 0.f ? dpa[j] * inv_keep : 0.f;
        dP[j][e] = dp;
        delta[j] += dp * fabsf(pv[j][e]);
      }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    // delta over the row: lanes with the same head group u (xor offsets LS, 2LS, .. 32), then waves
#pragma unroll
    for (int j = 0; j < HL; ++j) {
#pragma unroll
      for (int o = LS; o < 64; o <<= 1) delta[j] += __shfl_xor(delta[j], o, 64);
    }
    if constexpr (WPR > 1) {
      if (lane < LS) {
#pragma unroll
        for (int j = 0; j < HL; ++j) redd[wave][lane * HL + j] = delta[j];
      }
      __syncthreads();
      const int w0 = rsub * WPR;
#pragma unroll
      for (int j = 0; j < HL; ++j) {
        float a = 0.f;
        for (int q = 0; q < WPR; ++q) a += redd[w0 + q][u * HL + j];
        delta[j] = a;
      }
      __syncthreads();
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < HL; ++j) {
        vu_f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (jc + e < N) ? fabsf(pv[j][e]) * (dP[j][e] - delta[j]) * scale : 0.f;
        vu_st4(dA + off + j * hs, o);
      }
    }
  }
  // head-mix weight gradient: reduce over lanes with the same u, then over waves, then atomics.
  // aW[j][q] is in relative order: column q = k*HL+jj is head (u^k)*HL+jj.
#pragma unroll
  for (int j = 0; j < HL; ++j) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      float v = aW[j][q];
#pragma unroll
      for (int o = LS; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      if (lane < LS) {
        const int k = q / HL, jj = q % HL;
        red[wave][(lane * HL + j) * H + (lane ^ k) * HL + jj] = v;
      }
    }
    float v = ac[j];
#pragma unroll
    for (int o = LS; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    if (lane < LS) red[wave][H * H + lane * HL + j] = v;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < H * H + H; q += blockDim.x) {
    float v = 0.f;
    for (int w = 0; w < 16; ++w) v += red[w][q];
    if (q < H * H) atomicAdd(dW + q, v); else atomicAdd(dc + (q - H * H), v);
  }
}

template <typename T, int H>
int launch_map_bwd_row(const void* Ps, void* dA, const float* W, const float* c, const float* gamma, const float* stats,
                       float* dW, float* dc, int B, int N, int ld, float inv_keep, float scale, hipStream_t st) {
  constexpr int LS = H >= 8 ? 4 : (H >= 4 ? 2 : 1);
  const long long rows = (long long)B * N;
  const int need = (ld / 4) * LS;          // threads per row
#define VU_MBS(TPRv)                                                                                                   \
  {                                                                                                                    \
    long long grid = (rows + (1024 / TPRv) - 1) / (1024 / TPRv);                                                       \
    if (grid > 512) grid = 512;                                                                                        \
    hipLaunchKernelGGL((map_bwd_split_kernel<T, H, LS, TPRv>), dim3((unsigned)grid), dim3(1024), 0, st, (const T*)Ps, \
                       (T*)dA, W, c, gamma, stats, dW, dc, rows, N, ld, inv_keep, scale);                              \
  }
  if (need <= 64) VU_MBS(64)
  else if (need <= 256) VU_MBS(256)
  else if (need <= 1024) VU_MBS(1024)
  else return 1;
#undef VU_MBS
  if (vu_prof_on()) vu_prof_note("map_bwd_split_kernel", 0.0, (double)B * H * N * N * 3 * sizeof(T));
  return vu_check_launch("vu_map_bwd");
}

}  // namespace

int vu_k_map_bwd(int dtype, const void* Ps, void* dAhat_dS, const float* W, const float* c, const float* gamma,
                 const float* stats, float* dW, float* dc, int B, int H, int N, int ld, float inv_keep, float scale,
                 hipStream_t st) {
  const int ls = H >= 8 ? 4 : (H >= 4 ? 2 : 1);
  if ((ld / 4) * ls > 1024) return vu_k_map_bwd_2sweep(dtype, Ps, dAhat_dS, W, c, gamma, stats, dW, dc, B, H, N, ld, inv_keep, scale, st);
#define VU_MB(Tt, Hh) return launch_map_bwd_row<Tt, Hh>(Ps, dAhat_dS, W, c, gamma, stats, dW, dc, B, N, ld, inv_keep, scale, st)
  if (dtype == 0) { switch (H) { case 1: VU_MB(float, 1); case 2: VU_MB(float, 2); case 4: VU_MB(float, 4); case 8: VU_MB(float, 8); } }
  else { switch (H) { case 1: VU_MB(bf16_t, 1); case 2: VU_MB(bf16_t, 2); case 4: VU_MB(bf16_t, 4); case 8: VU_MB(bf16_t, 8); } }
#undef VU_MB
  vu_set_error("map_bwd: num_heads %d not supported", H);
  return VU_EUNSUPPORTED;
}

// =============================================================================================
// BatchNorm-backward statistics WITHOUT a pass over the maps.  With dAhat_g = dO_g v_g^T:
//   s1_g = sum dAhat_g                 = sum_b sum_t (sum_i dO_g[i,t]) (sum_j v_g[j,t])
//   r_g  = sum dAhat_g * Ahat_g        = sum_{b,i,t} dO_g[i,t] O_g[i,t]        (O = Ahat v, saved)
// and, because Ahat = gamma*xhat + beta,   s2_g = sum dAhat_g*xhat_g = (r_g - beta_g s1_g) / gamma_g.
// Reads three (B,N,D) tensors instead of two (B,h,N,N) maps.
// =============================================================================================
namespace {

// one block per sample: threads = (row lanes) x (4-feature vectors of the D-wide row), coalesced
// 8/16-byte loads; per-feature column sums are combined across row lanes in LDS, then reduced
// per head.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(const T* __restrict__ dO, const T* __restrict__ O,
                                                           const T* __restrict__ v, float* partials, int N, int D, int H) {
  __shared__ float sdo[256][4], sv[256][4];
  __shared__ float hs1[16], hr[16];
  const int b = blockIdx.x, d = D / H;
  const long long base = (long long)b * N * D;
  const int nvec = D >> 2;
  const int TV = nvec < 256 ? nvec : 256;     // feature vectors handled side by side
  const int RL = 256 / TV;                    // row lanes
  const int tv = threadIdx.x % TV, rl = threadIdx.x / TV;
  if (threadIdx.x < 16) { hs1[threadIdx.x] = 0.f; hr[threadIdx.x] = 0.f; }
  __syncthreads();
  for (int v0 = 0; v0 < nvec; v0 += TV) {
    const int vi = v0 + tv;
    const bool ok = rl < RL && vi < nvec;
    float cdo[4] = {0, 0, 0, 0}, cv[4] = {0, 0, 0, 0}, rr[4] = {0, 0, 0, 0};
    if (ok) {
      for (int i = rl; i < N; i += RL) {
        const long long o = base + (long long)i * D + vi * 4;
        const vu_f4 a = vu_ld4(dO + o), c = vu_ld4(v + o), q = vu_ld4(O + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) { cdo[e] += a.v[e]; cv[e] += c.v[e]; rr[e] += a.v[e] * q.v[e]; }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sdo[threadIdx.x][e] = cdo[e]; sv[threadIdx.x][e] = cv[e]; }
    // r needs no cross-lane pairing: add straight into the head bins
    if (ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&hr[(vi * 4 + e) / d], rr[e]);
    }
    __syncthreads();
    if (rl == 0 && vi < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = 0.f, c = 0.f;
        for (int q = 0; q < RL; ++q) { a += sdo[q * TV + tv][e]; c += sv[q * TV + tv][e]; }
        atomicAdd(&hs1[(vi * 4 + e) / d], a * c);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < H) {
    partials[(long long)b * 2 * H + threadIdx.x] = hs1[threadIdx.x];
    partials[(long long)b * 2 * H + H + threadIdx.x] = hr[threadIdx.x];
  }
}

// stats layout: Wf[H*H] cf[H] mean[H] rstd[H] m1[H] m2[H]
__global__ void bn_bwd_small_finalize_kernel(const float* partials, int nb, const float* gamma, const float* beta,
                                             float* stats, float* dgamma, float* dbeta, int H, double count, int training) {
  const int g = threadIdx.x;
  if (g >= H) return;
  double s1 = 0.0, r = 0.0;
  for (int i = 0; i < nb; ++i) { s1 += (double)partials[i * 2 * H + g]; r += (double)partials[i * 2 * H + H + g]; }
  const double gm = gamma[g];
  const double s2 = fabs(gm) > 1e-20 ? (r - (double)beta[g] * s1) / gm : 0.0;
  dbeta[g] += (float)s1;
  dgamma[g] += (float)s2;
  stats[H * H + 3 * H + g] = training ? (float)(s1 / count) : 0.f;
  stats[H * H + 4 * H + g] = training ? (float)(s2 / count) : 0.f;
}

}  // namespace

// partials: >= B*2*H floats
int vu_k_bn_bwd_small(int dtype, const void* dO, const void* O, const void* v, const float* gamma, const float* beta,
                      float* stats, float* dgamma, float* dbeta, float* partials, int B, int N, int D, int H,
                      int training, hipStream_t st) {
  if (dtype == 0) hipLaunchKernelGGL((bn_bwd_small_kernel<float>), dim3(B), dim3(256), 0, st, (const float*)dO, (const float*)O, (const float*)v, partials, N, D, H);
  else hipLaunchKernelGGL((bn_bwd_small_kernel<bf16_t>), dim3(B), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)O, (const bf16_t*)v, partials, N, D, H);
  hipLaunchKernelGGL(bn_bwd_small_finalize_kernel, dim3(1), dim3(64), 0, st, partials, B, gamma, beta, stats, dgamma, dbeta, H,
                     (double)B * N * N, training);
  if (vu_prof_on()) vu_prof_note("bn_bwd_small(2 kernels)", 0.0, 3.0 * B * N * D * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_bn_bwd_small");
}
