// Attention backward over the maps: map backward (MFMA / VALU forms), MFMA mix statistics, BatchNorm-backward
// statistics without a pass over the maps.  (Split from vu_attn.hip to keep the translation units parallel.)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "vu_kernels.h"
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
                                                          long long rows, int N, int ld, float inv_keep, float scale,
                                                          float* __restrict__ part) {
  constexpr int RPB = 256 / TPR;
  // W and the backward tables (written by bn_bwd_small_finalize_kernel) come through scalar loads
  const float* __restrict__ tX = stats + H * H + 5 * H;     // X[H*H], Xc[H], Gs[H]
  const float* __restrict__ tM = stats + H * H + 3 * H;     // m1[H], m2[H]
  __shared__ float redd[4][H];
  __shared__ float red[4][H * H + H];
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
          float xh = tX[H * H + g];
#pragma unroll
          for (int h = 0; h < H; ++h) xh += tX[g * H + h] * pt[h];
          dAg[g] = cv ? tX[H * H + H + g] * (dAh[g][e] - tM[g] - xh * tM[H + g]) : 0.f;
          ac[g] += dAg[g];
#pragma unroll
          for (int h = 0; h < H; ++h) aW[g * H + h] += dAg[g] * pt[h];
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float dp = 0.f;
#pragma unroll
          for (int g = 0; g < H; ++g) dp += W[g * H + h] * dAg[g];
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
    } else if constexpr (TPR == 16) {       // four rows per wave: sums over aligned 16-lane groups
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float v = delta[h];
#pragma unroll
        for (int m = 1; m <= 8; m <<= 1) v += __shfl_xor(v, m, 64);
        delta[h] = v;
      }
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
    if (part) part[(long long)blockIdx.x * (H * H + H) + i] = v;      // summed in block order by map_bwd_partials_reduce_kernel
    else if (i < H * H) atomicAdd(dW + i, v); else atomicAdd(dc + (i - H * H), v);
  }
}

// Deterministic tail of the map-backward kernels: every workgroup leaves its h*h + h head-mix gradient sums in a slab of
// the model workspace (vu_gemm_get_scratch) and one wave per output adds them in block order.  (The float atomics they
// replace all hit the same 72 addresses at the end of the kernel, ~50 ns each when contended: 512 workgroups spent
// half of the level-1 kernel's 52 us in that tail.)
__global__ __launch_bounds__(1024) void map_bwd_partials_reduce_kernel(const float* __restrict__ part, int nblocks, int n, int hh,
                                                                       float* dW, float* dc) {
  const int i = blockIdx.x * 16 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;          // wave-uniform
  float s[4] = {0.f, 0.f, 0.f, 0.f};      // four independent chains over loads issued together (fixed order all the same)
  int b = lane;
  for (; b + 3 * 64 < nblocks; b += 4 * 64) {
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = part[(long long)(b + 64 * u) * n + i];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += x[u];
  }
  for (; b < nblocks; b += 64) s[0] += part[(long long)b * n + i];
  float a = (s[0] + s[1]) + (s[2] + s[3]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
  if (lane == 0) { if (i < hh) dW[i] += a; else dc[i - hh] += a; }
}
// (nblocks rows of h*h + h floats; `defer`: the slab came from the executor's arena and the reduce is queued, vu_gemm.h)
inline float* map_bwd_slab(long long nblocks, int H, hipStream_t st, bool& defer) {
  const size_t floats = (size_t)nblocks * (H * H + H);
  float* p = vu_deferred_take(floats, st);
  defer = p != nullptr;
  if (defer) return p;
  void* scr = nullptr; size_t bytes = 0;
  vu_gemm_get_scratch(&scr, &bytes);
  return (scr && bytes >= floats * sizeof(float)) ? (float*)scr : nullptr;
}
inline void map_bwd_reduce(float* part, long long nblocks, int H, float* dW, float* dc, hipStream_t st, bool defer) {
  const int n = H * H + H;
  if (defer) {
    vu_defred d;
    memset(&d, 0, sizeof(d));
    d.kind = VU_DEFRED_MAP; d.nblocks = (int)nblocks; d.n = n; d.hh = H * H; d.part = part; d.dst[0] = dW; d.dst[1] = dc;
    vu_deferred_push(d);
    return;
  }
  hipLaunchKernelGGL(map_bwd_partials_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(1024), 0, st, part, (int)nblocks, n, H * H, dW, dc);
}

// ---------------------------------------------------------------------------------------------
// map_bwd_mfma_kernel (bf16 storage, one 256-thread block per row, ld <= 1024): as
// map_bwd_row_kernel, but the head-mix weight gradient dW[g,h] = sum_pos dA_g P~_h is taken off the
// VALU: every lane drops its dA and P~ values (bf16) into two [16][row] LDS images and the four
// waves contract them over the row's positions with v_mfma_f32_16x16x32_bf16 (a ones row in the
// P~ image yields dc = sum dA_g for free).  Without the 64 per-lane accumulators the kernel fits
// two waves per SIMD.
// ---------------------------------------------------------------------------------------------
template <int H, int NWV>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void map_bwd_mfma_kernel(const bf16_t* __restrict__ Ps, bf16_t* dA,
                                                              const float* __restrict__ W, const float* __restrict__ c,
                                                              const float* __restrict__ gamma, const float* __restrict__ stats,
                                                              float* dW, float* dc, long long rows, int N, int ld,
                                                              float inv_keep, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // W and the backward tables are wave-uniform and indexed with compile-time constants: they
  // arrive through scalar loads (SGPRs), not through LDS / VGPRs
  const float* __restrict__ tX0 = stats + H * H + 5 * H;    // X[H*H], Xc[H], Gs[H]
  const float* __restrict__ tM0 = stats + H * H + 3 * H;    // m1[H], m2[H]
  __shared__ float redd[NWV][H];
  __shared__ float red[NWV][256];
  const int ldk = (ld + 31) / 32 * 32;        // positions rounded up to whole k-steps
  const int LDP = ldk + 8;                    // image row stride (elements)
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw);   // [H][LDP]    rows g: dA_g, high bf16 part (MFMA rows >= H read as zero)
  bf16_t* sL = sA + H * LDP;                          // [H][LDP]    low part: dA = hi + lo keeps 16 significant bits
  bf16_t* sB = sL + H * LDP;                          // [H+1][LDP]  rows h: P~_h, row H: ones
  for (int i = threadIdx.x; i < (3 * H + 1) * LDP; i += blockDim.x) {
    const int r = i / LDP, col = i % LDP;
    sA[i] = (r == 3 * H && col < N) ? (bf16_t)1.0f : (bf16_t)0.0f;   // sB row H = ones over the valid positions
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int jc = threadIdx.x * 4;
  const long long hs = (long long)N * ld;
  const int nks = ldk / 32;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const bool live = jc < ld;
    const long long b = row / N;
    const int i = (int)(row - b * N);
    const long long off = (b * H * N + i) * (long long)ld + jc;
    // (measured: forcing the scalar table loads to stay inside the row loop is slower - 6.3 vs 4.4 ms
    // per step - than letting the compiler hoist them and spill part of the table to VGPR lanes)
    const float* tX = tX0; const float* tM = tM0; const float* Wt = W;
    float pv[H][4], dP[H][4], dAg[H][4], delta[H];
#pragma unroll
    for (int h = 0; h < H; ++h) delta[h] = 0.f;
    if (live) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const vu_f4 v = vu_ld4(Ps + off + h * hs);
        const vu_f4 d = vu_ld4(dA + off + h * hs);
#pragma unroll
        for (int e = 0; e < 4; ++e) { pv[h][e] = v.v[e]; dAg[h][e] = d.v[e]; }
      }
      // two phases so that only one 8x8 table (X, then W) is live in scalar registers at a time
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool cv = jc + e < N;
        float pt[H];
#pragma unroll
        for (int h = 0; h < H; ++h) pt[h] = pv[h][e] > 0.f ? pv[h][e] * inv_keep : 0.f;
#pragma unroll
        for (int g = 0; g < H; ++g) {
          float xh = tX[H * H + g];
#pragma unroll
          for (int h = 0; h < H; ++h) xh = fmaf(tX[g * H + h], pt[h], xh);
          dAg[g][e] = cv ? tX[H * H + H + g] * (dAg[g][e] - tM[g] - xh * tM[H + g]) : 0.f;
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float dp = 0.f;
#pragma unroll
          for (int g = 0; g < H; ++g) dp = fmaf(Wt[g * H + h], dAg[g][e], dp);
          dp = pv[h][e] > 0.f ? dp * inv_keep : 0.f;
          dP[h][e] = dp;
          delta[h] = fmaf(dp, fabsf(pv[h][e]), delta[h]);
        }
      }
      // LDS images for the MFMA contraction (4 consecutive positions per store)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        vu_f4 a4, l4, p4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a4.v[e] = dAg[h][e];
          l4.v[e] = dAg[h][e] - (float)(bf16_t)dAg[h][e];
          p4.v[e] = pv[h][e] > 0.f ? pv[h][e] * inv_keep : 0.f;
        }
        vu_st4(sA + h * LDP + jc, a4);
        vu_st4(sL + h * LDP + jc, l4);
        vu_st4(sB + h * LDP + jc, p4);
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float v = vu_wave_sum(delta[h]);
      if (lane == 0) redd[wave][h] = v;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < H; ++h) { float dsum = 0.f;
#pragma unroll
      for (int q = 0; q < NWV; ++q) dsum += redd[q][h];
      delta[h] = dsum; }
    // contraction over the row's positions, k-steps dealt round-robin to the 4 waves
    for (int ks = wave; ks < nks; ks += NWV) {
      const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
      const bf16x8 af = l15 < H ? *reinterpret_cast<const bf16x8*>(sA + l15 * LDP + ks * 32 + lg * 8) : zero8;
      const bf16x8 lf = l15 < H ? *reinterpret_cast<const bf16x8*>(sL + l15 * LDP + ks * 32 + lg * 8) : zero8;
      const bf16x8 bf = l15 <= H ? *reinterpret_cast<const bf16x8*>(sB + l15 * LDP + ks * 32 + lg * 8) : zero8;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, bf, acc, 0, 0, 0);
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
    __syncthreads();     // images and redd are rewritten by the next row
  }
  // acc: C[row g = lg*4 + r][col = l15]; columns < H are dW[g][h], column H is dc[g]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(lg * 4 + r) * 16 + l15] = acc[r];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int g = threadIdx.x / 16, hcol = threadIdx.x % 16;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < NWV; ++q) v += red[q][threadIdx.x];
    if (g < H && hcol < H) atomicAdd(dW + g * H + hcol, v);
    else if (g < H && hcol == H) atomicAdd(dc + g, v);
  }
}

// ---------------------------------------------------------------------------------------------
// map_bwd_mm_kernel (bf16 storage, H = 8, one 256-thread block per row, ld <= 1024): all three
// 8x8 head contractions of the map backward run on the matrix cores.
//   "own" layout    : thread t owns the position quad 4t..4t+3, all 8 heads  (B operand of MFMA #1)
//   "result" layout : what v_mfma_f32_16x16x32 returns when four position sets (one per 16-lane
//                     group) are stacked along K with a block-diagonal A operand: lane (l15, lg)
//                     gets heads 4*(lg&1)..+3 of the quads owned by lanes (l15, lg>>1) ["A"] and
//                     (l15, 2 + (lg>>1)) ["B"].
// MFMA #1: xhat pre-activation  = X  (8x8) . P~   -> result layout
// elementwise (result layout, dAhat / P loaded from HBM directly in that layout):  dA, later dP, dS
// MFMA #2: dP~ = W^T (8x8) . dA : the accumulators of #1's layout ARE its B operand (k-slot
//          (lg, 4m+r)), and its output lands in the same result layout - no lane movement.
// MFMA #3: dW[g,h] = sum_pos dA_g P~_h through two [head][position] LDS images (as
//          map_bwd_mfma_kernel), dA split hi/lo.
// ---------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// zero the bf16 halves whose sign bit is set (the dropped probabilities): packed signed-16 max with 0
__device__ __forceinline__ unsigned keep_pos(unsigned w) {
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}
__device__ __forceinline__ float half_f(unsigned w, int odd) { return __uint_as_float(odd ? (w & 0xffff0000u) : (w << 16)); }
__device__ __forceinline__ float unpk(const uint2& q, int e) { return half_f(e < 2 ? q.x : q.y, e & 1); }
__device__ __forceinline__ unsigned pk2(float a, float b) {
  const bf16x2v v = {(bf16_t)a, (bf16_t)b};
  return __builtin_bit_cast(unsigned, v);
}
// word made of the low (odd = 0) or high (odd = 1) bf16 halves of x (-> low half) and y (-> high half)
__device__ __forceinline__ unsigned halves(unsigned x, unsigned y, int odd) {
  return __builtin_amdgcn_perm(y, x, odd ? 0x07060302u : 0x05040100u);
}
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16_t)v[j];
  return r;
}

// EXACT: N % 4 == 0, position quads are wholly valid or wholly padding.
// WPR = waves per map row: 4 (256 < ld <= 1024: the block's four waves share a row) or 1 (ld <= 256: every
// wave owns a row of its own - its LDS images, its delta reduction and its dW contraction - no barriers).
template <bool EXACT, int WPR>
__global__ __launch_bounds__(256, 2) void map_bwd_mm_kernel(const bf16_t* __restrict__ Ps, bf16_t* dA,
                                                            const float* __restrict__ W, const float* __restrict__ c,
                                                            const float* __restrict__ gamma, const float* __restrict__ stats,
                                                            float* dW, float* dc, long long rows, int N, int ld,
                                                            float inv_keep, float scale, float* __restrict__ part) {
  constexpr int H = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ float redd[4][H];
  __shared__ float red[4][256];
  const int ldk = (ld + 31) / 32 * 32;
  const int LDP = ldk + 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wsub = WPR == 4 ? wave : 0;               // wave's index within its row
  bf16_t* sbase = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* sA = sbase + (WPR == 4 ? 0 : wave) * (3 * H + 1) * LDP;   // [H][LDP]   dA hi
  bf16_t* sL = sA + H * LDP;                          // [H][LDP]   dA lo
  bf16_t* sB = sL + H * LDP;                          // [H+1][LDP] kept P (without 1/keep), row H = ones
  for (int i = threadIdx.x; i < (WPR == 4 ? 1 : 4) * (3 * H + 1) * LDP; i += blockDim.x) {
    const int r = (i / LDP) % (3 * H + 1), col = i % LDP;
    sbase[i] = (r == 3 * H && col < N) ? (bf16_t)1.0f : (bf16_t)0.0f;
  }
  const int l15 = lane & 15, lg = lane >> 4;
  const float* tX = stats + H * H + 5 * H;     // X[H*H] = W*rstd_g, Xc[H], Gs[H]
  const float* tM = stats + H * H + 3 * H;     // m1[H], m2[H]
  // ---- constant A operands (block diagonal over the four lane-group position sets) ------------
  // #1, MFMA m: row16 = l15 -> (q' = 2m + l15/8, g = l15%8); k-slot (lg, j): X[g][j]/keep if lg == q'
  // #2, MFMA m2: row16 = l15 -> (q'' = 2m2 + l15/8, h = l15%8); k-slot (lg, j = 4m + r) is
  //     (q' = 2m + lg/2, g = 4(lg&1) + r): W[g][h] scale/keep if q' == q''
  bf16x8 A1[2], A2[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float v1[8], v2[8];
    const int qrow = 2 * m + (l15 >> 3), hr = l15 & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v1[j] = (lg == qrow) ? tX[hr * H + j] * inv_keep : 0.f;
      const int mk = j >> 2, r = j & 3;
      const int qk = 2 * mk + (lg >> 1), gk = 4 * (lg & 1) + r;
      v2[j] = (qk == qrow) ? W[gk * H + hr] * (inv_keep * scale) : 0.f;
    }
    A1[m] = pack8(v1);
    A2[m] = pack8(v2);
  }
  const int hbase = 4 * (lg & 1);
  // dA = Gs (dAhat - m1 - (acc + Xc) m2) = Gs dAhat + K1 + K2 acc
  float Gs4[4], K1[4], K2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gs = tX[H * H + H + hbase + r], xc = tX[H * H + hbase + r];
    const float m1 = tM[hbase + r], m2 = tM[H + hbase + r];
    Gs4[r] = gs; K1[r] = -gs * (m1 + xc * m2); K2[r] = -gs * m2;
  }
  __syncthreads();
  const unsigned hs = (unsigned)N * (unsigned)ld;     // launcher guarantees 8 * N * ld < 2^31
  const int nks = ldk / 32;
  const int nquads = EXACT ? (N >> 2) : (ld >> 2);
  const int qown = WPR == 4 ? threadIdx.x : lane;         // own quad
  const int qA = 64 * wsub + 16 * (lg >> 1) + l15;        // result-layout quads
  const int qB = qA + 32;
  f32x4 accw = {0.f, 0.f, 0.f, 0.f};

  uint2 pown[H], PA[4], PB[4], QA[4], QB[4];             // packed bf16 quads: P (own / result layout), dAhat
  auto load_row = [&](long long row, uint2 (&po)[H], uint2 (&pa)[4], uint2 (&pb)[4], uint2 (&qa)[4], uint2 (&qb)[4]) {
    const long long b = row / N;
    const int i = (int)(row - b * N);
    const long long base = (b * H * N + i) * (long long)ld;
    const bf16_t* __restrict__ Prow = Ps + base;
    const bf16_t* Drow = dA + base;
#pragma unroll
    for (int h = 0; h < H; ++h) po[h] = make_uint2(0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { pa[r] = make_uint2(0, 0); pb[r] = make_uint2(0, 0); qa[r] = make_uint2(0, 0); qb[r] = make_uint2(0, 0); }
    if (qown < nquads) {
#pragma unroll
      for (int h = 0; h < H; ++h) po[h] = *reinterpret_cast<const uint2*>(Prow + (h * hs + 4u * qown));
    }
    if (qA < nquads) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned o = (hbase + r) * hs + 4u * qA;
        pa[r] = *reinterpret_cast<const uint2*>(Prow + o);
        qa[r] = *reinterpret_cast<const uint2*>(Drow + o);
      }
    }
    if (qB < nquads) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned o = (hbase + r) * hs + 4u * qB;
        pb[r] = *reinterpret_cast<const uint2*>(Prow + o);
        qb[r] = *reinterpret_cast<const uint2*>(Drow + o);
      }
    }
  };
  const long long row0 = WPR == 4 ? (long long)blockIdx.x : (long long)blockIdx.x * 4 + wave;
  const long long rstep = WPR == 4 ? (long long)gridDim.x : (long long)gridDim.x * 4;
  if (row0 < rows) load_row(row0, pown, PA, PB, QA, QB);
  for (long long row = row0; row < rows; row += rstep) {
    // ---- own layout: kept probabilities, P image, B operands of #1 -------------------------------
    unsigned b1w[4][4];
#pragma unroll
    for (int h = 0; h < H; ++h) { pown[h].x = keep_pos(pown[h].x); pown[h].y = keep_pos(pown[h].y); }
    if (qown < nquads) {
#pragma unroll
      for (int h = 0; h < H; ++h) *reinterpret_cast<uint2*>(sB + h * LDP + 4 * qown) = pown[h];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b1w[0][j] = halves(pown[2 * j].x, pown[2 * j + 1].x, 0);
      b1w[1][j] = halves(pown[2 * j].x, pown[2 * j + 1].x, 1);
      b1w[2][j] = halves(pown[2 * j].y, pown[2 * j + 1].y, 0);
      b1w[3][j] = halves(pown[2 * j].y, pown[2 * j + 1].y, 1);
    }
    unsigned hiw[4][4], low[4][4];        // [e][word]: bf16 pairs (r0,r1),(r2,r3) of quad A, then of quad B
    float dPa[4][4], dPb[4][4], delta[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x4 b1u = {b1w[e][0], b1w[e][1], b1w[e][2], b1w[e][3]};
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, b1u);
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[0], b1, c0, 0, 0, 0);   // quads of lane groups 0,1
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[1], b1, c1, 0, 0, 0);   // quads of lane groups 2,3
      float va[4], vb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        va[r] = fmaf(K2[r], c0[r], fmaf(Gs4[r], unpk(QA[r], e), K1[r]));
        vb[r] = fmaf(K2[r], c1[r], fmaf(Gs4[r], unpk(QB[r], e), K1[r]));
        if (!EXACT) { va[r] = (4 * qA + e < N) ? va[r] : 0.f; vb[r] = (4 * qB + e < N) ? vb[r] : 0.f; }
      }
      hiw[e][0] = pk2(va[0], va[1]); hiw[e][1] = pk2(va[2], va[3]);
      hiw[e][2] = pk2(vb[0], vb[1]); hiw[e][3] = pk2(vb[2], vb[3]);
      low[e][0] = pk2(va[0] - half_f(hiw[e][0], 0), va[1] - half_f(hiw[e][0], 1));
      low[e][1] = pk2(va[2] - half_f(hiw[e][1], 0), va[3] - half_f(hiw[e][1], 1));
      low[e][2] = pk2(vb[0] - half_f(hiw[e][2], 0), vb[1] - half_f(hiw[e][2], 1));
      low[e][3] = pk2(vb[2] - half_f(hiw[e][3], 0), vb[3] - half_f(hiw[e][3], 1));
      const u32x4 b2u = {hiw[e][0], hiw[e][1], hiw[e][2], hiw[e][3]};
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, b2u);
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[0], b2, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[1], b2, d1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pa = unpk(PA[r], e), pb = unpk(PB[r], e);
        const float xa = pa > 0.f ? d0[r] : 0.f;        // = dP scale (1/keep and scale sit in A2)
        const float xb = pb > 0.f ? d1[r] : 0.f;
        dPa[r][e] = xa; dPb[r][e] = xb;
        delta[r] = fmaf(xa, fabsf(pa), fmaf(xb, fabsf(pb), delta[r]));
      }
    }
    // ---- LDS images of dA for dW (result layout -> [head][position]) ----------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int wv = r >> 1, od = r & 1;
      if (qA < nquads) {
        *reinterpret_cast<uint2*>(sA + (hbase + r) * LDP + 4 * qA) =
            make_uint2(halves(hiw[0][wv], hiw[1][wv], od), halves(hiw[2][wv], hiw[3][wv], od));
        *reinterpret_cast<uint2*>(sL + (hbase + r) * LDP + 4 * qA) =
            make_uint2(halves(low[0][wv], low[1][wv], od), halves(low[2][wv], low[3][wv], od));
      }
      if (qB < nquads) {
        *reinterpret_cast<uint2*>(sA + (hbase + r) * LDP + 4 * qB) =
            make_uint2(halves(hiw[0][2 + wv], hiw[1][2 + wv], od), halves(hiw[2][2 + wv], hiw[3][2 + wv], od));
        *reinterpret_cast<uint2*>(sL + (hbase + r) * LDP + 4 * qB) =
            make_uint2(halves(low[0][2 + wv], low[1][2 + wv], od), halves(low[2][2 + wv], low[3][2 + wv], od));
      }
    }
    // ---- next row's loads fly during the reduction / contraction / store phase --------------------
    uint2 nP[4], nPB[4];
    const long long nrow = row + rstep;
    if (nrow < rows) load_row(nrow, pown, nP, nPB, QA, QB);
    // ---- delta_h over the row: lanes with the same (lg & 1) hold the same heads ----------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = delta[r];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 32, 64);
      if (WPR == 4) { if (l15 == 0 && lg < 2) redd[wave][4 * lg + r] = v; }
      else delta[r] = v;                 // one wave = one row: the shuffles already hold the row sum
    }
    if (WPR == 4) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) delta[r] = redd[0][hbase + r] + redd[1][hbase + r] + redd[2][hbase + r] + redd[3][hbase + r];
    }
    for (int ks = wsub; ks < nks; ks += WPR) {
      const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
      const bf16x8 af = l15 < H ? *reinterpret_cast<const bf16x8*>(sA + l15 * LDP + ks * 32 + lg * 8) : zero8;
      const bf16x8 lf = l15 < H ? *reinterpret_cast<const bf16x8*>(sL + l15 * LDP + ks * 32 + lg * 8) : zero8;
      const bf16x8 bf = l15 <= H ? *reinterpret_cast<const bf16x8*>(sB + l15 * LDP + ks * 32 + lg * 8) : zero8;
      accw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, accw, 0, 0, 0);
      accw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, bf, accw, 0, 0, 0);
    }
    // ---- dS = |p| (dP - delta) scale, result layout, 4 consecutive positions per store ----------
    {
      const long long b = row / N;
      const int i = (int)(row - b * N);
      bf16_t* Drow = dA + (b * H * N + i) * (long long)ld;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned o = (hbase + r) * hs;
        float oa[4], ob[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          oa[e] = fabsf(unpk(PA[r], e)) * (dPa[r][e] - delta[r]);
          ob[e] = fabsf(unpk(PB[r], e)) * (dPb[r][e] - delta[r]);
          if (!EXACT) { oa[e] = (4 * qA + e < N) ? oa[e] : 0.f; ob[e] = (4 * qB + e < N) ? ob[e] : 0.f; }
        }
        if (qA < nquads) *reinterpret_cast<uint2*>(Drow + (o + 4u * qA)) = make_uint2(pk2(oa[0], oa[1]), pk2(oa[2], oa[3]));
        if (qB < nquads) *reinterpret_cast<uint2*>(Drow + (o + 4u * qB)) = make_uint2(pk2(ob[0], ob[1]), pk2(ob[2], ob[3]));
      }
      if (EXACT && qown >= nquads && 4 * qown < ld) {      // the padding quad of the row
#pragma unroll
        for (int h = 0; h < H; ++h) *reinterpret_cast<uint2*>(Drow + (h * hs + 4u * qown)) = make_uint2(0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { PA[r] = nP[r]; PB[r] = nPB[r]; }
    if (WPR == 4) __syncthreads();       // (WPR = 1: a wave's LDS traffic is ordered by itself)
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(lg * 4 + r) * 16 + l15] = accw[r];
  __syncthreads();
  {
    const int g = threadIdx.x / 16, hcol = threadIdx.x % 16;
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (part) {
      if (g < H && hcol < H) part[(long long)blockIdx.x * (H * H + H) + g * H + hcol] = v * inv_keep;
      else if (g < H && hcol == H) part[(long long)blockIdx.x * (H * H + H) + H * H + g] = v;
    }
    else if (g < H && hcol < H) atomicAdd(dW + g * H + hcol, v * inv_keep);    // the P image holds kept p, not p/keep
    else if (g < H && hcol == H) atomicAdd(dc + g, v);
  }
}

// ---------------------------------------------------------------------------------------------
// map_bwd_mm_long_kernel (bf16 storage, H = 8, ld > 1024: the 512x512 inputs, N = 4096): the MFMA
// stages of map_bwd_mm_kernel applied to CHUNKS of 1024 columns of a row, one 256-thread block per
// row, in two sweeps - the softmax backward needs delta_h = sum_j dP_h P_h over the WHOLE row:
//   sweep 1, per chunk: xhat (MFMA #1) -> dA -> dP (MFMA #2) -> delta partial; dA / P images -> dW (MFMA #3)
//   sweep 2, per chunk: the same two MFMAs again -> dS = |p| (dP - delta) scale, written over dAhat
// (5 map passes instead of 3; the second read of a row comes right after the first).  The loads of
// the next chunk fly during the reduction / contraction / store phase of the current one.
// ---------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ __launch_bounds__(256, 2) void map_bwd_mm_long_kernel(const bf16_t* __restrict__ Ps, bf16_t* dA,
                                                                 const float* __restrict__ W, const float* __restrict__ c,
                                                                 const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                 float* dW, float* dc, long long rows, int N, int ld,
                                                                 float inv_keep, float scale) {
  constexpr int H = 8, CW = 1024, CQ = CW / 4, LDP = CW + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ float redd[4][H];
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw);   // [H][LDP]   dA hi
  bf16_t* sL = sA + H * LDP;                          // [H][LDP]   dA lo
  bf16_t* sB = sL + H * LDP;                          // [H+1][LDP] kept P (without 1/keep), row H = ones over the valid columns
  for (int i = threadIdx.x; i < (3 * H + 1) * LDP; i += blockDim.x) sA[i] = (bf16_t)0.0f;
  const int l15 = lane & 15, lg = lane >> 4;
  const float* tX = stats + H * H + 5 * H;     // X[H*H] = W*rstd_g, Xc[H], Gs[H]
  const float* tM = stats + H * H + 3 * H;     // m1[H], m2[H]
  bf16x8 A1[2], A2[2];                         // constant A operands: see map_bwd_mm_kernel
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float v1[8], v2[8];
    const int qrow = 2 * m + (l15 >> 3), hr = l15 & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v1[j] = (lg == qrow) ? tX[hr * H + j] * inv_keep : 0.f;
      const int mk = j >> 2, r = j & 3;
      const int qk = 2 * mk + (lg >> 1), gk = 4 * (lg & 1) + r;
      v2[j] = (qk == qrow) ? W[gk * H + hr] * (inv_keep * scale) : 0.f;
    }
    A1[m] = pack8(v1);
    A2[m] = pack8(v2);
  }
  const int hbase = 4 * (lg & 1);
  float Gs4[4], K1[4], K2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gs = tX[H * H + H + hbase + r], xc = tX[H * H + hbase + r];
    const float m1 = tM[hbase + r], m2 = tM[H + hbase + r];
    Gs4[r] = gs; K1[r] = -gs * (m1 + xc * m2); K2[r] = -gs * m2;
  }
  __syncthreads();
  const unsigned hs = (unsigned)N * (unsigned)ld;     // launcher guarantees 8 * N * ld < 2^31
  const int nquads = EXACT ? (N >> 2) : (ld >> 2);    // quads of a row that are loaded
  const int nch = (ld + CW - 1) / CW;                 // chunks per row
  const int qown = threadIdx.x;                       // own quad within the chunk
  const int qA = 64 * wave + 16 * (lg >> 1) + l15;    // result-layout quads within the chunk
  const int qB = qA + 32;
  f32x4 accw = {0.f, 0.f, 0.f, 0.f};

  uint2 pown[H], PA[4], PB[4], QA[4], QB[4];
  auto load_chunk = [&](long long row, int ch, uint2 (&po)[H], uint2 (&pa)[4], uint2 (&pb)[4], uint2 (&qa)[4], uint2 (&qb)[4]) {
    const long long b = row / N;
    const int i = (int)(row - b * N);
    const long long base = (b * H * N + i) * (long long)ld + (long long)ch * CW;
    const bf16_t* __restrict__ Prow = Ps + base;
    const bf16_t* Drow = dA + base;
    const int nq = nquads - ch * CQ;                  // quads of this chunk (may exceed CQ: the q's are < CQ)
#pragma unroll
    for (int h = 0; h < H; ++h) po[h] = make_uint2(0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { pa[r] = make_uint2(0, 0); pb[r] = make_uint2(0, 0); qa[r] = make_uint2(0, 0); qb[r] = make_uint2(0, 0); }
    if (qown < nq) {
#pragma unroll
      for (int h = 0; h < H; ++h) po[h] = *reinterpret_cast<const uint2*>(Prow + (h * hs + 4u * qown));
    }
    if (qA < nq) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned o = (hbase + r) * hs + 4u * qA;
        pa[r] = *reinterpret_cast<const uint2*>(Prow + o);
        qa[r] = *reinterpret_cast<const uint2*>(Drow + o);
      }
    }
    if (qB < nq) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned o = (hbase + r) * hs + 4u * qB;
        pb[r] = *reinterpret_cast<const uint2*>(Prow + o);
        qb[r] = *reinterpret_cast<const uint2*>(Drow + o);
      }
    }
  };
  // work items in order: (row, sweep, chunk)
  long long row = blockIdx.x;
  int sweep = 0, ch = 0;
  float dtot[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) load_chunk(row, 0, pown, PA, PB, QA, QB);
  while (row < rows) {
    const int c0 = ch * CW;                           // first column of the chunk
    const int nq = nquads - ch * CQ;
    // ---- own layout: kept probabilities, P image (sweep 1 only), B operands of #1 ----------------
    unsigned b1w[4][4];
#pragma unroll
    for (int h = 0; h < H; ++h) { pown[h].x = keep_pos(pown[h].x); pown[h].y = keep_pos(pown[h].y); }
    if (sweep == 0) {
#pragma unroll
      for (int h = 0; h < H; ++h) *reinterpret_cast<uint2*>(sB + h * LDP + 4 * qown) = pown[h];   // zero where not loaded
      const int cq = c0 + 4 * qown;
      const unsigned w0 = (cq < N ? 0x3f80u : 0u) | (cq + 1 < N ? 0x3f800000u : 0u);
      const unsigned w1 = (cq + 2 < N ? 0x3f80u : 0u) | (cq + 3 < N ? 0x3f800000u : 0u);
      *reinterpret_cast<uint2*>(sB + H * LDP + 4 * qown) = make_uint2(w0, w1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b1w[0][j] = halves(pown[2 * j].x, pown[2 * j + 1].x, 0);
      b1w[1][j] = halves(pown[2 * j].x, pown[2 * j + 1].x, 1);
      b1w[2][j] = halves(pown[2 * j].y, pown[2 * j + 1].y, 0);
      b1w[3][j] = halves(pown[2 * j].y, pown[2 * j + 1].y, 1);
    }
    unsigned hiw[4][4], low[4][4];
    float dPa[4][4], dPb[4][4], delta[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x4 b1u = {b1w[e][0], b1w[e][1], b1w[e][2], b1w[e][3]};
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, b1u);
      f32x4 c0v = {0.f, 0.f, 0.f, 0.f}, c1v = {0.f, 0.f, 0.f, 0.f};
      c0v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[0], b1, c0v, 0, 0, 0);
      c1v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[1], b1, c1v, 0, 0, 0);
      float va[4], vb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        va[r] = fmaf(K2[r], c0v[r], fmaf(Gs4[r], unpk(QA[r], e), K1[r]));
        vb[r] = fmaf(K2[r], c1v[r], fmaf(Gs4[r], unpk(QB[r], e), K1[r]));
        if (!EXACT) { va[r] = (c0 + 4 * qA + e < N) ? va[r] : 0.f; vb[r] = (c0 + 4 * qB + e < N) ? vb[r] : 0.f; }
      }
      hiw[e][0] = pk2(va[0], va[1]); hiw[e][1] = pk2(va[2], va[3]);
      hiw[e][2] = pk2(vb[0], vb[1]); hiw[e][3] = pk2(vb[2], vb[3]);
      if (sweep == 0) {
        low[e][0] = pk2(va[0] - half_f(hiw[e][0], 0), va[1] - half_f(hiw[e][0], 1));
        low[e][1] = pk2(va[2] - half_f(hiw[e][1], 0), va[3] - half_f(hiw[e][1], 1));
        low[e][2] = pk2(vb[0] - half_f(hiw[e][2], 0), vb[1] - half_f(hiw[e][2], 1));
        low[e][3] = pk2(vb[2] - half_f(hiw[e][3], 0), vb[3] - half_f(hiw[e][3], 1));
      }
      const u32x4 b2u = {hiw[e][0], hiw[e][1], hiw[e][2], hiw[e][3]};
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, b2u);
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[0], b2, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[1], b2, d1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pa = unpk(PA[r], e), pb = unpk(PB[r], e);
        const float xa = pa > 0.f ? d0[r] : 0.f;        // = dP scale (1/keep and scale sit in A2)
        const float xb = pb > 0.f ? d1[r] : 0.f;
        dPa[r][e] = xa; dPb[r][e] = xb;
        delta[r] = fmaf(xa, fabsf(pa), fmaf(xb, fabsf(pb), delta[r]));
      }
    }
    if (sweep == 0) {
      // ---- LDS images of dA for dW (result layout -> [head][position]); zero where the chunk has no quad
      const bool okA = qA < nq, okB = qB < nq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int wv = r >> 1, od = r & 1;
        *reinterpret_cast<uint2*>(sA + (hbase + r) * LDP + 4 * qA) =
            okA ? make_uint2(halves(hiw[0][wv], hiw[1][wv], od), halves(hiw[2][wv], hiw[3][wv], od)) : make_uint2(0, 0);
        *reinterpret_cast<uint2*>(sL + (hbase + r) * LDP + 4 * qA) =
            okA ? make_uint2(halves(low[0][wv], low[1][wv], od), halves(low[2][wv], low[3][wv], od)) : make_uint2(0, 0);
        *reinterpret_cast<uint2*>(sA + (hbase + r) * LDP + 4 * qB) =
            okB ? make_uint2(halves(hiw[0][2 + wv], hiw[1][2 + wv], od), halves(hiw[2][2 + wv], hiw[3][2 + wv], od)) : make_uint2(0, 0);
        *reinterpret_cast<uint2*>(sL + (hbase + r) * LDP + 4 * qB) =
            okB ? make_uint2(halves(low[0][2 + wv], low[1][2 + wv], od), halves(low[2][2 + wv], low[3][2 + wv], od)) : make_uint2(0, 0);
      }
    }
    // ---- the next item's loads fly during the reduction / contraction / store phase ----------------
    long long nrow = row; int nsweep = sweep, nchk = ch + 1;
    if (nchk == nch) { nchk = 0; if (++nsweep == 2) { nsweep = 0; nrow = row + gridDim.x; } }
    uint2 nP[4], nPB[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { nP[r] = make_uint2(0, 0); nPB[r] = make_uint2(0, 0); }
    if (nrow < rows) load_chunk(nrow, nchk, pown, nP, nPB, QA, QB);
    if (sweep == 0) {
      // ---- delta_h over the chunk, added to the row's total --------------------------------------
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = delta[r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 32, 64);
        if (l15 == 0 && lg < 2) redd[wave][4 * lg + r] = v;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) dtot[r] += redd[0][hbase + r] + redd[1][hbase + r] + redd[2][hbase + r] + redd[3][hbase + r];
      for (int ks = wave; ks < CW / 32; ks += 4) {
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const bf16x8 af = l15 < H ? *reinterpret_cast<const bf16x8*>(sA + l15 * LDP + ks * 32 + lg * 8) : zero8;
        const bf16x8 lf = l15 < H ? *reinterpret_cast<const bf16x8*>(sL + l15 * LDP + ks * 32 + lg * 8) : zero8;
        const bf16x8 bf = l15 <= H ? *reinterpret_cast<const bf16x8*>(sB + l15 * LDP + ks * 32 + lg * 8) : zero8;
        accw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, accw, 0, 0, 0);
        accw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, bf, accw, 0, 0, 0);
      }
      __syncthreads();     // images and redd are rewritten by the next chunk
    } else {
      // ---- dS = |p| (dP - delta) scale, result layout, 4 consecutive positions per store ----------
      const long long b = row / N;
      const int i = (int)(row - b * N);
      bf16_t* Drow = dA + (b * H * N + i) * (long long)ld + c0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned o = (hbase + r) * hs;
        float oa[4], ob[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          oa[e] = fabsf(unpk(PA[r], e)) * (dPa[r][e] - dtot[r]);
          ob[e] = fabsf(unpk(PB[r], e)) * (dPb[r][e] - dtot[r]);
          if (!EXACT) { oa[e] = (c0 + 4 * qA + e < N) ? oa[e] : 0.f; ob[e] = (c0 + 4 * qB + e < N) ? ob[e] : 0.f; }
        }
        if (qA < nq) *reinterpret_cast<uint2*>(Drow + (o + 4u * qA)) = make_uint2(pk2(oa[0], oa[1]), pk2(oa[2], oa[3]));
        if (qB < nq) *reinterpret_cast<uint2*>(Drow + (o + 4u * qB)) = make_uint2(pk2(ob[0], ob[1]), pk2(ob[2], ob[3]));
      }
      if (EXACT && qown >= nq && c0 + 4 * qown < ld) {      // the padding quad of the row
#pragma unroll
        for (int h = 0; h < H; ++h) *reinterpret_cast<uint2*>(Drow + (h * hs + 4u * qown)) = make_uint2(0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { PA[r] = nP[r]; PB[r] = nPB[r]; }
    if (nsweep == 0 && nchk == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dtot[r] = 0.f;
    }
    row = nrow; sweep = nsweep; ch = nchk;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(lg * 4 + r) * 16 + l15] = accw[r];
  __syncthreads();
  {
    const int g = threadIdx.x / 16, hcol = threadIdx.x % 16;
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (g < H && hcol < H) atomicAdd(dW + g * H + hcol, v * inv_keep);    // the P image holds kept p, not p/keep
    else if (g < H && hcol == H) atomicAdd(dc + g, v);
  }
}

// ---------------------------------------------------------------------------------------------
// mix_stats_mm_kernel (bf16 storage, H = 8, 256 < ld <= 1024): the BatchNorm batch statistics of
// the mixed maps, s1_g = sum (a_g - shift_g), s2_g = sum (a_g - shift_g)^2 with
// a_g = sum_h W[g,h] P~_h, one 256-thread block per map row.  The 8x8 mix runs on the matrix cores
// exactly as MFMA #1 of map_bwd_mm_kernel (own-layout B operand, block-diagonal A, result layout
// out); W/keep enters as a bf16 hi + lo pair (16 significant bits) and -shift_g as the accumulator
// input, so that the VALU only squares and sums.
// ---------------------------------------------------------------------------------------------
// STORE: also write the centred mixed map  Ac_g = a_g - shift_g  (bf16).  BatchNorm is affine in it,
// Ahat_g = sc_g Ac_g + kappa_g, and so are the two products that consume Ahat (O = Ahat v, dv = Ahat^T dO):
// they take Ac and apply (sc, kappa) with a column sum in their epilogue - the normalised map is never
// written and the separate apply pass (one more read of P) disappears.
template <bool EXACT, bool STORE>
__global__ __launch_bounds__(256) void mix_stats_mm_kernel(const bf16_t* __restrict__ Ps, const float* __restrict__ W,
                                                           float* __restrict__ partials, bf16_t* __restrict__ Ac,
                                                           long long rows, int N, int ld, float inv_keep) {
  constexpr int H = 8;
  __shared__ float red[4][2 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  bf16x8 Ah[2], Al[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float vh[8], vl[8];
    const int qrow = 2 * m + (l15 >> 3), hr = l15 & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (lg == qrow) ? W[hr * H + j] * inv_keep : 0.f;
      vh[j] = (float)(bf16_t)v;
      vl[j] = v - vh[j];
    }
    Ah[m] = pack8(vh);
    Al[m] = pack8(vl);
  }
  const int hbase = 4 * (lg & 1);
  f32x4 cin;       // -shift_g, shift_g = sum_h W[g,h] / N  (the exact mean without dropout; the bias cancels)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float sacc = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) sacc += W[(hbase + r) * H + h];
    cin[r] = -sacc / (float)N;
  }
  const unsigned hs = (unsigned)N * (unsigned)ld;
  const int nquads = EXACT ? (N >> 2) : (ld >> 2);
  const int qown = threadIdx.x;
  const int qA = 64 * wave + 16 * (lg >> 1) + l15, qB = qA + 32;
  float mA[4], mB[4];       // validity of the result-layout positions (1 / 0)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mA[e] = (EXACT ? qA < nquads : 4 * qA + e < N) ? 1.f : 0.f;
    mB[e] = (EXACT ? qB < nquads : 4 * qB + e < N) ? 1.f : 0.f;
  }
  float s1a[4] = {0.f, 0.f, 0.f, 0.f}, s2a[4] = {0.f, 0.f, 0.f, 0.f}, s1b[4] = {0.f, 0.f, 0.f, 0.f}, s2b[4] = {0.f, 0.f, 0.f, 0.f};
  uint2 pown[H];
  auto load_row = [&](long long row) {
    const long long b = row / N;
    const int i = (int)(row - b * N);
    const bf16_t* __restrict__ Prow = Ps + (b * H * N + i) * (long long)ld;
#pragma unroll
    for (int h = 0; h < H; ++h) pown[h] = make_uint2(0, 0);
    if (qown < nquads) {
#pragma unroll
      for (int h = 0; h < H; ++h) pown[h] = *reinterpret_cast<const uint2*>(Prow + (h * hs + 4u * qown));
    }
  };
  if ((long long)blockIdx.x < rows) load_row(blockIdx.x);
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    unsigned b1w[4][4];
#pragma unroll
    for (int h = 0; h < H; ++h) { pown[h].x = keep_pos(pown[h].x); pown[h].y = keep_pos(pown[h].y); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b1w[0][j] = halves(pown[2 * j].x, pown[2 * j + 1].x, 0);
      b1w[1][j] = halves(pown[2 * j].x, pown[2 * j + 1].x, 1);
      b1w[2][j] = halves(pown[2 * j].y, pown[2 * j + 1].y, 0);
      b1w[3][j] = halves(pown[2 * j].y, pown[2 * j + 1].y, 1);
    }
    const long long nrow = row + gridDim.x;
    if (nrow < rows) load_row(nrow);          // the next row is in flight during the MFMAs
    float oa[4][4], ob[4][4];                 // [head r][element e] of the two result-layout quads (STORE)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x4 b1u = {b1w[e][0], b1w[e][1], b1w[e][2], b1w[e][3]};
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, b1u);
      f32x4 c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[0], b1, cin, 0, 0, 0);
      f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[1], b1, cin, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[0], b1, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[1], b1, c1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (EXACT) {
          s1a[r] += c0[r]; s2a[r] = fmaf(c0[r], c0[r], s2a[r]);
          s1b[r] += c1[r]; s2b[r] = fmaf(c1[r], c1[r], s2b[r]);
          if constexpr (STORE) { oa[r][e] = c0[r]; ob[r][e] = c1[r]; }
        } else {
          const float a = c0[r] * mA[e], bq = c1[r] * mB[e];
          s1a[r] += a; s2a[r] = fmaf(a, a, s2a[r]);
          s1b[r] += bq; s2b[r] = fmaf(bq, bq, s2b[r]);
          if constexpr (STORE) { oa[r][e] = a; ob[r][e] = bq; }
        }
      }
    }
    if constexpr (STORE) {
      const long long b = row / N;
      const int i = (int)(row - b * N);
      bf16_t* Arow = Ac + (b * H * N + i) * (long long)ld;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned o = (hbase + r) * hs;
        if (qA < nquads) *reinterpret_cast<uint2*>(Arow + (o + 4u * qA)) = make_uint2(pk2(oa[r][0], oa[r][1]), pk2(oa[r][2], oa[r][3]));
        if (qB < nquads) *reinterpret_cast<uint2*>(Arow + (o + 4u * qB)) = make_uint2(pk2(ob[r][0], ob[r][1]), pk2(ob[r][2], ob[r][3]));
      }
      if (EXACT && qown >= nquads && 4 * qown < ld) {      // the padding quad of the row: zeros (consumers clamp, never skip)
#pragma unroll
        for (int h = 0; h < H; ++h) *reinterpret_cast<uint2*>(Arow + (h * hs + 4u * qown)) = make_uint2(0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // (EXACT: a lane's two quads are valid or padding for every row - drop the padding sums here)
    float v1 = EXACT ? s1a[r] * mA[0] + s1b[r] * mB[0] : s1a[r] + s1b[r];
    float v2 = EXACT ? s2a[r] * mA[0] + s2b[r] * mB[0] : s2a[r] + s2b[r];
#pragma unroll
    for (int m = 1; m <= 8; m <<= 1) { v1 += __shfl_xor(v1, m, 64); v2 += __shfl_xor(v2, m, 64); }
    v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64);
    if (l15 == 0 && lg < 2) { red[wave][4 * lg + r] = v1; red[wave][H + 4 * lg + r] = v2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * H)
    partials[blockIdx.x * 2 * H + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// A/B switch for measurements: VU_MAP_BWD_VALU=1 keeps the VALU map-backward kernels (read once)
inline bool map_bwd_valu_forced() { static const bool v = getenv("VU_MAP_BWD_VALU") != nullptr; return v; }

template <int WPR>
int launch_map_bwd_mm(const void* Ps, void* dA, const float* W, const float* c, const float* gamma, const float* stats,
                      float* dW, float* dc, int B, int N, int ld, float inv_keep, float scale, hipStream_t st) {
  const long long rows = (long long)B * N;
  const int ldk = (ld + 31) / 32 * 32;
  const size_t lds = (size_t)(WPR == 4 ? 1 : 4) * (3 * 8 + 1) * (ldk + 8) * 2;
  auto kern = (N % 4 == 0) ? map_bwd_mm_kernel<true, WPR> : map_bwd_mm_kernel<false, WPR>;
  if (lds > 40 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { vu_set_error("map_bwd: cannot reserve %zu bytes of LDS", lds); return VU_ELAUNCH; }
  }
  long long grid = WPR == 4 ? rows : (rows + 3) / 4;
  const long long cap = WPR == 4 ? 1024 : 512;     // (measured: 512 blocks of 4 one-wave rows beat 1024 at N = 196)
  static const int cap_env = getenv("VU_MAP_BWD_CAP") ? atoi(getenv("VU_MAP_BWD_CAP")) : 0;     // measurement switch
  if (cap_env > 0 && WPR == 1) { if (grid > cap_env) grid = cap_env; }
  else if (grid > cap) grid = cap;
  bool defer = false;
  float* part = map_bwd_slab(grid, 8, st, defer);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, (const bf16_t*)Ps, (bf16_t*)dA, W, c, gamma, stats, dW, dc,
                     rows, N, ld, inv_keep, scale, part);
  if (part) map_bwd_reduce(part, grid, 8, dW, dc, st, defer);
  if (vu_prof_on()) vu_prof_note(WPR == 4 ? "map_bwd_mm_kernel" : "map_bwd_mm_kernel<1 wave/row>", 0.0, (double)B * 8 * N * N * 3 * 2.0);
  return vu_check_launch("vu_map_bwd");
}

int launch_map_bwd_mm_long(const void* Ps, void* dA, const float* W, const float* c, const float* gamma, const float* stats,
                           float* dW, float* dc, int B, int N, int ld, float inv_keep, float scale, hipStream_t st) {
  const long long rows = (long long)B * N;
  const size_t lds = (size_t)(3 * 8 + 1) * (1024 + 8) * 2;
  auto kern = (N % 4 == 0) ? map_bwd_mm_long_kernel<true> : map_bwd_mm_long_kernel<false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { vu_set_error("map_bwd: cannot reserve %zu bytes of LDS", lds); return VU_ELAUNCH; }
  long long grid = rows; if (grid > 512) grid = 512;     // two blocks per CU (236 VGPRs, 51.6 KB of images each), persistent
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, (const bf16_t*)Ps, (bf16_t*)dA, W, c, gamma, stats, dW, dc,
                     rows, N, ld, inv_keep, scale);
  if (vu_prof_on()) vu_prof_note("map_bwd_mm_long_kernel", 0.0, (double)B * 8 * N * N * 3 * 2.0);
  return vu_check_launch("vu_map_bwd");
}

template <typename T, int H>
int launch_map_bwd_row(const void* Ps, void* dA, const float* W, const float* c, const float* gamma, const float* stats,
                       float* dW, float* dc, int B, int N, int ld, float inv_keep, float scale, hipStream_t st) {
  const long long rows = (long long)B * N;
  if constexpr (H == 8 && sizeof(T) == 2) {
    if (ld <= 256 && ld >= 64 && !map_bwd_valu_forced())      // (round 6: level 0, ld = 56, measured on this kernel too: 27 vs 34 us per launch, step unchanged within noise - not taken)
      return launch_map_bwd_mm<1>(Ps, dA, W, c, gamma, stats, dW, dc, B, N, ld, inv_keep, scale, st);
  }
  if (ld <= 64) {
    // short rows (level 0: N = 49): 16 threads per row, and few workgroups - every workgroup ends in h*h + h float atomics
    // on the same addresses, ~50 ns each when contended (784 four-row workgroups took 57 us for 2.8 MB of map)
    long long grid = (rows + 15) / 16;
    bool defer = false;
    float* part = map_bwd_slab(grid > 512 ? 512 : grid, H, st, defer);
    const long long cap = part ? 512 : 96;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((map_bwd_row_kernel<T, H, 16>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)Ps, (T*)dA, W, c, gamma,
                       stats, dW, dc, rows, N, ld, inv_keep, scale, part);
    if (part) map_bwd_reduce(part, grid, H, dW, dc, st, defer);
    if (vu_prof_on()) vu_prof_note("map_bwd_row_kernel", 0.0, (double)B * H * N * N * 3 * sizeof(T));
  } else if (ld <= 256) {
    long long grid = (rows + 3) / 4; if (grid > 2048) grid = 2048;
    bool defer = false;
    float* part = map_bwd_slab(grid, H, st, defer);
    hipLaunchKernelGGL((map_bwd_row_kernel<T, H, 64>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)Ps, (T*)dA, W, c, gamma,
                       stats, dW, dc, rows, N, ld, inv_keep, scale, part);
    if (part) map_bwd_reduce(part, grid, H, dW, dc, st, defer);
    if (vu_prof_on()) vu_prof_note("map_bwd_row_kernel", 0.0, (double)B * H * N * N * 3 * sizeof(T));
  } else if (sizeof(T) == 2 && (ld <= 1024 || (ld <= 4096 && H <= 4))) {
    // one block per row: 256 threads (ld <= 1024) or 1024 threads (ld <= 4096; 128-VGPR budget -> H <= 4)
    const int ldk = (ld + 31) / 32 * 32;
    const size_t lds = (size_t)(3 * H + 1) * (ldk + 8) * 2;
    const bool big = ld > 1024;
    if constexpr (H == 8) {
      if (!big && !map_bwd_valu_forced()) return launch_map_bwd_mm<4>(Ps, dA, W, c, gamma, stats, dW, dc, B, N, ld, inv_keep, scale, st);
    }
    auto kern = big ? map_bwd_mfma_kernel<(H <= 4 ? H : 4), 16> : map_bwd_mfma_kernel<H, 4>;
    if (lds > 40 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { vu_set_error("map_bwd: cannot reserve %zu bytes of LDS", lds); return VU_ELAUNCH; }
    }
    long long grid = rows; if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(big ? 1024 : 256), lds, st, (const bf16_t*)Ps, (bf16_t*)dA, W, c, gamma,
                       stats, dW, dc, rows, N, ld, inv_keep, scale);
    if (vu_prof_on()) vu_prof_note("map_bwd_mfma_kernel", 0.0, (double)B * H * N * N * 3 * sizeof(T));
  } else if (ld > 1024) {
    return 1;   // caller falls back to the two-sweep kernel
  } else {
    long long grid = rows; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL((map_bwd_row_kernel<T, H, 256>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)Ps, (T*)dA, W, c, gamma,
                       stats, dW, dc, rows, N, ld, inv_keep, scale, (float*)nullptr);
    if (vu_prof_on()) vu_prof_note("map_bwd_row_kernel", 0.0, (double)B * H * N * N * 3 * sizeof(T));
  }
  return vu_check_launch("vu_map_bwd");
}

}  // namespace

int vu_k_map_bwd(int dtype, const void* Ps, void* dAhat_dS, const float* W, const float* c, const float* gamma,
                 const float* stats, float* dW, float* dc, int B, int H, int N, int ld, float inv_keep, float scale,
                 hipStream_t st) {
  if (dtype == 1 && H == 8 && ld > 1024 && 8LL * N * ld < 2147483647LL && !map_bwd_valu_forced())
    return launch_map_bwd_mm_long(Ps, dAhat_dS, W, c, gamma, stats, dW, dc, B, N, ld, inv_keep, scale, st);
  // rows > 1024 with <= 4 heads (Lite level 2): the two-sweep kernel (two waves per SIMD, loads one trip ahead) beats the
  // one-row-per-1024-thread-block MFMA form, 4.3 vs 4.9 ms per launch at N = 3136; VU_MAP_BWD_ONE_SWEEP=1 restores the latter
  static const bool one_sweep = getenv("VU_MAP_BWD_ONE_SWEEP") != nullptr;
  if (ld > 4096 || (ld > 1024 && (dtype == 0 || H > 4 || !one_sweep)))
    return vu_k_map_bwd_2sweep(dtype, Ps, dAhat_dS, W, c, gamma, stats, dW, dc, B, H, N, ld, inv_keep, scale, st);
#define VU_MB(Tt, Hh) return launch_map_bwd_row<Tt, Hh>(Ps, dAhat_dS, W, c, gamma, stats, dW, dc, B, N, ld, inv_keep, scale, st)
  if (dtype == 0) { switch (H) { case 1: VU_MB(float, 1); case 2: VU_MB(float, 2); case 4: VU_MB(float, 4); case 8: VU_MB(float, 8); } }
  else { switch (H) { case 1: VU_MB(bf16_t, 1); case 2: VU_MB(bf16_t, 2); case 4: VU_MB(bf16_t, 4); case 8: VU_MB(bf16_t, 8); } }
#undef VU_MB
  vu_set_error("map_bwd: num_heads %d not supported", H);
  return VU_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// mix_center_small_kernel (bf16 storage, H = 8, short rows: ld <= 256 - Base / Large level 1, N = 196, and level 0, N = 49): the
// same two results as mix_stats_mm_kernel<., true> - shifted batch moments and the centred mixed map Ac - for rows too short to
// fill a 256-thread block.  One thread owns a quad of 4 keys of one map row in all 8 heads (8 x 8 bytes in, 8 x 8 bytes out) and
// mixes them on the VALU (64 multiply-adds per map position: 0.3 GFLOP per launch at 64 images - the kernel is its two map
// passes); threads run over (row, quad) pairs, consecutive threads over the consecutive quads of a row.  Round 6: with it levels
// 1 / 0 take the centred-map form too (one pass that reads P and writes Ac instead of mix_stats + mix_apply: two reads, one write).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mix_center_small_kernel(const bf16_t* __restrict__ Ps, const float* __restrict__ W,
                                                               float* __restrict__ partials, bf16_t* __restrict__ Ac,
                                                               long long rows, int N, int ld, float inv_keep) {
  constexpr int H = 8;
  __shared__ float red[4][2 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The 64 coefficients live in VECTOR registers, fetched once through an address the compiler cannot prove uniform: as wave-uniform
  // data they would be scalar loads that hipcc may re-issue inside the loop under register pressure - the pattern that returned
  // wrong data beside concurrent kernels in conv_fwd_kernel.
  int vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  float Wk[H][H], cin[H];
#pragma unroll
  for (int g = 0; g < H; ++g) {
    float sacc = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) { const float w = W[g * H + h + vz]; Wk[g][h] = w * inv_keep; sacc += w; }
    cin[g] = -sacc / (float)N;      // -shift_g, shift_g = sum_h W[g,h] / N (the exact mean without dropout; the bias cancels)
  }
  const int nq = ld >> 2;
  const unsigned hs = (unsigned)N * (unsigned)ld;
  const long long total = rows * nq;
  float s1[H], s2[H];
#pragma unroll
  for (int g = 0; g < H; ++g) { s1[g] = 0.f; s2[g] = 0.f; }
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long row = t / nq;
    const int qd = (int)(t - row * nq);
    const long long b = row / N;
    const int i = (int)(row - b * N);
    const long long base = (b * H * N + i) * (long long)ld + 4 * qd;
    uint2 p[H];
#pragma unroll
    for (int h = 0; h < H; ++h) p[h] = *reinterpret_cast<const uint2*>(Ps + base + (long long)h * hs);
    float pv[H][4];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const unsigned x = keep_pos(p[h].x), y = keep_pos(p[h].y);
      pv[h][0] = half_f(x, 0); pv[h][1] = half_f(x, 1); pv[h][2] = half_f(y, 0); pv[h][3] = half_f(y, 1);
    }
#pragma unroll
    for (int g = 0; g < H; ++g) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = cin[g];
#pragma unroll
        for (int h = 0; h < H; ++h) a = fmaf(Wk[g][h], pv[h][e], a);
        a = (4 * qd + e < N) ? a : 0.f;            // padding columns: zeros in the map, nothing in the sums
        o[e] = a;
        s1[g] += a; s2[g] = fmaf(a, a, s2[g]);
      }
      *reinterpret_cast<uint2*>(Ac + base + (long long)g * hs) = make_uint2(pk2(o[0], o[1]), pk2(o[2], o[3]));
    }
  }
#pragma unroll
  for (int g = 0; g < H; ++g) {
    float v1 = s1[g], v2 = s2[g];
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) { v1 += __shfl_xor(v1, m, 64); v2 += __shfl_xor(v2, m, 64); }
    if (lane == 0) { red[wave][g] = v1; red[wave][H + g] = v2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * H)
    partials[blockIdx.x * 2 * H + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// returns VU_OK, a negative error, or 1 when the shape is not covered (the caller uses mix_stats_kernel).
// Ac != null: also write the centred mixed map (see mix_stats_mm_kernel).
int vu_k_mix_stats_mm(int dtype, const void* Ps, const float* W, float* partials, void* Ac, int nblocks, int B, int H, int N,
                      int ld, float inv_keep, hipStream_t st) {
  if (dtype != 1 || H != 8 || ld > 1024 || ld % 8 != 0 || (long long)H * N * ld >= 2147483647LL) return 1;
  const long long rows = (long long)B * N;
  if (ld <= 256) {        // short rows: the VALU kernel above (centred map only; the plain statistics stay on mix_stats_kernel)
    if (!Ac) return 1;
    hipLaunchKernelGGL(mix_center_small_kernel, dim3(nblocks), dim3(256), 0, st, (const bf16_t*)Ps, W, partials, (bf16_t*)Ac, rows, N, ld, inv_keep);
    if (vu_prof_on()) vu_prof_note("mix_center_small_kernel", 0.0, (double)B * H * N * N * 4.0);
    return vu_check_launch("vu_mix_stats_mm");
  }
  const bool ex = N % 4 == 0;
  auto kern = Ac ? (ex ? mix_stats_mm_kernel<true, true> : mix_stats_mm_kernel<false, true>)
                 : (ex ? mix_stats_mm_kernel<true, false> : mix_stats_mm_kernel<false, false>);
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), 0, st, (const bf16_t*)Ps, W, partials, (bf16_t*)Ac, rows, N, ld, inv_keep);
  if (vu_prof_on()) vu_prof_note(Ac ? "mix_center_mm_kernel" : "mix_stats_mm_kernel", 0.0, (double)B * H * N * N * (Ac ? 4.0 : 2.0));
  return vu_check_launch("vu_mix_stats_mm");
}

// =============================================================================================
// BatchNorm-backward statistics WITHOUT a pass over the maps.  With dAhat_g = dO_g v_g^T:
//   s1_g = sum dAhat_g                 = sum_b sum_t (sum_i dO_g[i,t]) (sum_j v_g[j,t])
//   r_g  = sum dAhat_g * Ahat_g        = sum_{b,i,t} dO_g[i,t] O_g[i,t]        (O = Ahat v, saved)
// and, because Ahat = gamma*xhat + beta,   s2_g = sum dAhat_g*xhat_g = (r_g - beta_g s1_g) / gamma_g.
// Reads three (B,N,D) tensors instead of two (B,h,N,N) maps.
// =============================================================================================
namespace {

// one block per (sample, head): feature columns side by side, row lanes stacked, LDS combine
// NT threads per workgroup: 256; 1024 (bf16 16-byte path only, VU_BN_BWD_WIDE=1: every thread makes ONE batch of 4 row loads) is a
// measurement variant.  Round 5: the row lanes of a feature octet are combined by wave shuffles and a parallel LDS sum instead of a
// serial loop of TV threads over RL LDS rows: 22.7 -> 17.5 us per call at 16 images, 28.2 -> 24.1 at 64 (tools/step_tags.py).
template <typename T, int NT>
__global__ __launch_bounds__(NT) void bn_bwd_small_kernel(const T* __restrict__ dO, const T* __restrict__ O,
                                                           const T* __restrict__ v, float* partials, int N, int D, int H) {
  __shared__ float sm[16];
  __shared__ float sdo[4 * NT], sv[4 * NT];
  const int b = blockIdx.x, g = blockIdx.y, d = D / H;
  const long long base = (long long)b * N * D + g * d;
  float s1 = 0.f, r = 0.f;
  if (sizeof(T) == 2 && d % 8 == 0 && d <= 2048) {
    // bf16, 16-byte loads: a thread owns 8 consecutive features of the rows i = rl, rl + RL, ... with 4 rows (12 loads,
    // 192 bytes) in flight - the rate of this kernel is bytes in flight per CU over the memory latency
    const int nv = d >> 3;
    int TV = NT;                        // feature octets side by side (power of two >= min(nv, NT))
    while (TV / 2 >= nv) TV /= 2;
    const int RL = NT / TV;
    const int tc = threadIdx.x % TV, rl = threadIdx.x / TV, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the row lanes of a feature octet meet in two steps: inside a wave by shuffles (TV < 64: lanes tc, tc + TV, ...), then one
    // value per wave (or per row lane when an octet row spans whole waves) through LDS, summed by 8 TV threads in parallel
    const bool leader = TV >= 64 || lane < TV;
    const int slot = TV >= 64 ? rl : wave, nslot = TV >= 64 ? RL : NT / 64;
    for (int v0 = 0; v0 < nv; v0 += TV) {
      const int vq = v0 + tc;
      float cdo[8], cv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { cdo[e] = 0.f; cv[e] = 0.f; }
      if (vq < nv) {
        const long long cb = base + 8 * vq;
        union U8 { uint4 u; bf16_t h[8]; };
        for (int i0 = rl; i0 < N; i0 += 4 * RL) {
          U8 a[4], vv[4], o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * RL;
            const long long off = cb + (long long)(i < N ? i : i0) * D;
            a[k].u = *reinterpret_cast<const uint4*>((const bf16_t*)dO + off);
            vv[k].u = *reinterpret_cast<const uint4*>((const bf16_t*)v + off);
            o[k].u = *reinterpret_cast<const uint4*>((const bf16_t*)O + off);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (i0 + k * RL < N) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float av = (float)a[k].h[e];
                cdo[e] += av; cv[e] += (float)vv[k].h[e]; r = fmaf(av, (float)o[k].h[e], r);
              }
            }
          }
        }
      }
      // column sums over the row lanes, then sum_t cdo[t] * cv[t]
      for (int m = TV; m < 64; m <<= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { cdo[e] += __shfl_xor(cdo[e], m, 64); cv[e] += __shfl_xor(cv[e], m, 64); }
      }
#pragma unroll
      for (int e0 = 0; e0 < 8; e0 += 4) {              // (two halves: 4 NT floats per array)
        if (leader) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { sdo[(slot * 4 + e) * TV + tc] = cdo[e0 + e]; sv[(slot * 4 + e) * TV + tc] = cv[e0 + e]; }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 4 * TV; idx += NT) {
          const int e = idx / TV, t = idx - e * TV;
          float a2 = 0.f, c2 = 0.f;
          for (int q = 0; q < nslot; ++q) { a2 += sdo[(q * 4 + e) * TV + t]; c2 += sv[(q * 4 + e) * TV + t]; }
          s1 = fmaf(a2, c2, s1);
        }
        __syncthreads();
      }
    }
  } else if (d % 4 == 0) {
    // a thread owns 4 consecutive features (one vector load per tensor and row) of the rows i = rl, rl + RL, ...
    const int nq = d >> 2;
    int TQ = 256;                       // feature quads handled side by side (power of two >= min(nq, 256))
    while (TQ / 2 >= nq) TQ /= 2;
    const int RL = 256 / TQ;
    const int tc = threadIdx.x % TQ, rl = threadIdx.x / TQ;
    for (int q0 = 0; q0 < nq; q0 += TQ) {
      const int qd = q0 + tc;
      float cdo[4] = {0.f, 0.f, 0.f, 0.f}, cv[4] = {0.f, 0.f, 0.f, 0.f};
      if (qd < nq) {
        const long long cb = base + 4 * qd;
#pragma unroll 4
        for (int i = rl; i < N; i += RL) {
          const vu_f4 a = vu_ld4(dO + cb + (long long)i * D), vv = vu_ld4(v + cb + (long long)i * D), o = vu_ld4(O + cb + (long long)i * D);
#pragma unroll
          for (int e = 0; e < 4; ++e) { cdo[e] += a.v[e]; cv[e] += vv.v[e]; r = fmaf(a.v[e], o.v[e], r); }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { sdo[e * 256 + threadIdx.x] = cdo[e]; sv[e * 256 + threadIdx.x] = cv[e]; }
      __syncthreads();
      if (rl == 0 && qd < nq) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = 0.f, c = 0.f;
          for (int q = 0; q < RL; ++q) { a += sdo[e * 256 + q * TQ + tc]; c += sv[e * 256 + q * TQ + tc]; }
          s1 = fmaf(a, c, s1);
        }
      }
      __syncthreads();
    }
  } else {
    int TCOL = 256;                       // feature columns handled side by side (power of two >= min(d,256))
    while (TCOL / 2 >= d) TCOL /= 2;
    const int RL = 256 / TCOL;            // row lanes per column
    const int tc = threadIdx.x % TCOL, rl = threadIdx.x / TCOL;
    for (int t0 = 0; t0 < d; t0 += TCOL) {
      const int t = t0 + tc;
      float cdo = 0.f, cv = 0.f;
      if (t < d) {
        for (int i = rl; i < N; i += RL) {
          const float a = vu_ld(dO + base + (long long)i * D + t);
          cdo += a;
          cv += vu_ld(v + base + (long long)i * D + t);
          r += a * vu_ld(O + base + (long long)i * D + t);
        }
      }
      sdo[threadIdx.x] = cdo; sv[threadIdx.x] = cv;
      __syncthreads();
      if (rl == 0 && t < d) {
        float a = 0.f, c = 0.f;
        for (int q = 0; q < RL; ++q) { a += sdo[q * TCOL + tc]; c += sv[q * TCOL + tc]; }
        s1 += a * c;
      }
      __syncthreads();
    }
  }
  s1 = vu_block_sum(s1, sm);
  r = vu_block_sum(r, sm);
  if (threadIdx.x == 0) {
    partials[(long long)b * 2 * H + g] = s1;
    partials[(long long)b * 2 * H + H + g] = r;
  }
}

// stats layout: Wf[H*H] cf[H] mean[H] rstd[H] m1[H] m2[H]
__global__ void bn_bwd_small_finalize_kernel(const float* partials, int nb, const float* gamma, const float* beta,
                                             const float* W, const float* c, float* stats, float* dgamma, float* dbeta,
                                             int H, double count, int training) {
  // 32 lanes per head (launched with 32 * H threads, H <= 16): strided partial sums, shuffle reduction
  const int g = threadIdx.x >> 5, sub = threadIdx.x & 31;
  if (g >= H) return;
  double s1 = 0.0, r = 0.0;
  for (int i = sub; i < nb; i += 32) { s1 += (double)partials[i * 2 * H + g]; r += (double)partials[i * 2 * H + H + g]; }
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); r += __shfl_xor(r, m, 64); }
  if (sub != 0) return;
  {  // tables read by the map-backward kernels through scalar loads
    const float rstd = stats[H * H + 2 * H + g];
    float* X = stats + H * H + 5 * H;
    for (int h = 0; h < H; ++h) X[g * H + h] = W[g * H + h] * rstd;
    X[H * H + g] = (c[g] - stats[H * H + H + g]) * rstd;
    X[H * H + H + g] = gamma[g] * rstd;
  }
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
                      const float* W, const float* c, float* stats, float* dgamma, float* dbeta, float* partials, int B, int N, int D, int H,
                      int training, hipStream_t st) {
  static const int wide_force = [] { const char* e = getenv("VU_BN_BWD_WIDE"); return e ? atoi(e) : -1; }();      // A/B switch: 0 / 1
  const int dh = D / H;
  const bool wide = dtype == 1 && dh % 8 == 0 && dh <= 2048 && wide_force > 0;      // (1024 threads: one batch of row loads per thread; measured level at 16 images - 17.9 against 17.5 us - and slower at 64)
  if (dtype == 0) hipLaunchKernelGGL((bn_bwd_small_kernel<float, 256>), dim3(B, H), dim3(256), 0, st, (const float*)dO, (const float*)O, (const float*)v, partials, N, D, H);
  else if (wide) hipLaunchKernelGGL((bn_bwd_small_kernel<bf16_t, 1024>), dim3(B, H), dim3(1024), 0, st, (const bf16_t*)dO, (const bf16_t*)O, (const bf16_t*)v, partials, N, D, H);
  else hipLaunchKernelGGL((bn_bwd_small_kernel<bf16_t, 256>), dim3(B, H), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)O, (const bf16_t*)v, partials, N, D, H);
  hipLaunchKernelGGL(bn_bwd_small_finalize_kernel, dim3(1), dim3(32 * H), 0, st, partials, B, gamma, beta, W, c, stats, dgamma, dbeta, H,
                     (double)B * N * N, training);
  if (vu_prof_on()) vu_prof_note("bn_bwd_small(2 kernels)", 0.0, 3.0 * B * N * D * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_bn_bwd_small");
}

