// Re-attention at SHORT rows as two fused kernels per direction (round 6): Base / Large level 1 (N = 196, 8 heads of 96 features)
// and level 0 (N = 49, 8 heads of 384).  model.py:155-161.
//
// The maps of these levels are small (N = 196: 39 MB per pass at 64 images) and the round-1 pipeline - scores -> mix + moments ->
// finalize -> PV, and backwards dO v^T -> BatchNorm / mix / softmax backward -> three products - ran eleven launches per module,
// each a fraction of the chip's rate (1.6 ms and 88 launches of the 11.3 ms Base step).  What couples the work is the HEAD MIX: every
// map position needs all 8 heads, while a logits tile wants one head per wave.  Both kernels here resolve that inside one workgroup:
//
//   phase 1  wave = head.  The wave's 16 QT queries against all keys of ITS head on the matrix cores, operands straight from global
//            memory (a lane's 8 k-slots = one 16-byte load of a q / k row; the 7 workgroups of an image run on one XCD, so K comes
//            from that L2), the row softmax / dropout (forward) in the accumulators, and the tile goes to LDS as [head][query][key].
//   phase 2  wave = 4 rows of the tile, ALL heads: 8 x 8 bytes from LDS per key quad, the 8 x 8 mix on the VALU, the per-head sums,
//            and the stores leave row-contiguously.
//
// forward  (attn_f1_kernel):  q, k -> P~ (sign-tagged bf16, the same values and the same dropout words as attn_scores_kernel) and the
//            centred mixed map Ac = W P~ / keep - shift, plus the shifted batch moments per workgroup (the layout bn_finalize_kernel
//            reads): scores + mix_center in one launch, the logits never leave the chip and P~ is written once instead of written,
//            read, and (round 1) read again.
#include <stdlib.h>
#include "vu_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned af_u4;

__device__ __forceinline__ f32x4 af_mma(const af_u4 a, const af_u4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned af_pk2(float a, float b) {
  const unsigned short lo = __builtin_bit_cast(unsigned short, (bf16_t)a), hi = __builtin_bit_cast(unsigned short, (bf16_t)b);
  return (unsigned)lo | ((unsigned)hi << 16);
}
__device__ __forceinline__ float af_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float af_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// zero the bf16 halves whose sign bit is set (dropped probabilities)
__device__ __forceinline__ unsigned af_keep_pos(unsigned w) {
  typedef short s2 __attribute__((ext_vector_type(2)));
  const s2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, w), z));
}

struct f1_args {
  const bf16_t *q, *k;
  bf16_t *Ps, *Ac;
  const float* W;          // (8, 8) head-mix weights
  float* partials;         // [gridDim.x][16]: sum (a_g - shift_g), sum (a_g - shift_g)^2
  int B, N, D, ld, nqt, nitems;
  float scale, inv_keep;
  vu_rng rng;
};

// NT: key tiles (16 NT >= N), KS: k-steps of 32 features per head (d = 32 KS), QT: query tiles of 16 per workgroup item
template <int NT, int KS, int QT>
__global__ __launch_bounds__(512) void attn_f1_kernel(const f1_args a) {
  constexpr int H = 8, QR = 16 * QT, LP = NT * 16 + 8;      // rows per item; LDS row pitch in elements (16-byte multiple, +8: bank spread)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* PL = reinterpret_cast<bf16_t*>(smem_raw);          // [H][QR][LP]
  __shared__ float red[8][16];
  const vu_rng rng = vu_rng_resolve(a.rng);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15c = lane & 15, gc = lane >> 4;
  const int N = a.N, D = a.D, ld = a.ld, d = 32 * KS;
  float s1[H], s2[H];
#pragma unroll
  for (int gg = 0; gg < H; ++gg) { s1[gg] = 0.f; s2[gg] = 0.f; }

  for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
    // the tiles of an image on ONE XCD (workgroups go to the XCDs round-robin): logical item = (item % 8) * (nitems / 8) + item / 8
    const int L = (a.nitems & 7) == 0 ? (item & 7) * (a.nitems >> 3) + (item >> 3) : item;
    const int b = L / a.nqt, qt = L - b * a.nqt;
    const int i0 = qt * QR;
    // (opaque copies of the lane coordinates, per item: otherwise every per-tile row address / hash index is hoisted out of the
    // item loop as a loop invariant - ~100 live registers, spilled: the same cure as in attn_scores_kernel)
    int l15 = l15c, g = gc;
    asm volatile("" : "+v"(l15), "+v"(g));
    // ================= phase 1: wave = head =================
    {
      const int h = wave;
      const bf16_t* qb = a.q + (long long)b * N * D + h * d;
      const bf16_t* kb = a.k + (long long)b * N * D + h * d;
      af_u4 qf[QT][KS];
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const int row = min(i0 + 16 * t + l15, N - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[t][ks] = *reinterpret_cast<const af_u4*>(qb + (long long)row * D + ks * 32 + g * 8);
      }
      f32x4 acc[QT][NT];
      // K fragments one key tile ahead (two register sets; the scheduling barrier keeps hipcc from hoisting all NT tiles' loads -
      // 156 registers at NT = 13 - above the products: the first version spilled 416 bytes per lane)
      af_u4 kf[2][KS];
      auto load_k = [&](int nt, af_u4 (&f)[KS]) {
        const int key = min(nt * 16 + l15, N - 1);           // keys >= N: any real row, masked below
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[ks] = *reinterpret_cast<const af_u4*>(kb + (long long)key * D + ks * 32 + g * 8);
      };
      load_k(0, kf[0]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt + 1 < NT) load_k(nt + 1, kf[(nt + 1) & 1]);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) acc[t][nt] = af_mma(kf[nt & 1][ks], qf[t][ks], acc[t][nt]);      // S^T = K Q^T: lane = query l15, keys 16 nt + 4 g + r
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const int i = i0 + 16 * t + l15;
        // ---- row softmax of the logits rounded to bf16 (as attn_scores_kernel: the unfused path stores them) ----
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sv = (float)(bf16_t)(acc[t][nt][r] * a.scale);
            sv = (nt * 16 + 4 * g + r < N) ? sv : -INFINITY;
            acc[t][nt][r] = sv;
            mx = fmaxf(mx, sv);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxl = mx * 1.44269504088896341f;
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(acc[t][nt][r], 1.44269504088896341f, -mxl));
            acc[t][nt][r] = e;
            sum += e;
          }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        // ---- dropout (the mask word of element (row, j) = hash of row ld + j, 16 bits per element) + sign tag -> LDS ----
        const uint32_t ib32 = (uint32_t)(((uint64_t)(b * H + h) * N + (uint64_t)min(i, N - 1)) * (uint64_t)ld);
        const uint32_t thr = rng.thr;
        bf16_t* prow = PL + ((long long)h * QR + 16 * t + l15) * LP;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j0 = nt * 16 + 4 * g;
          float o0 = acc[t][nt][0] * inv, o1 = acc[t][nt][1] * inv, o2 = acc[t][nt][2] * inv, o3 = acc[t][nt][3] * inv;
          if (thr) {
            const uint32_t wa = vu_hash_word32(rng, ib32 + j0), wb = vu_hash_word32(rng, ib32 + j0 + 2);
            o0 = __uint_as_float(__float_as_uint(o0) ^ (((wa & 0xffffu) - thr) & 0x80000000u));
            o1 = __uint_as_float(__float_as_uint(o1) ^ (((wa >> 16) - thr) & 0x80000000u));
            o2 = __uint_as_float(__float_as_uint(o2) ^ (((wb & 0xffffu) - thr) & 0x80000000u));
            o3 = __uint_as_float(__float_as_uint(o3) ^ (((wb >> 16) - thr) & 0x80000000u));
          }
          const bool rok = i < N;            // rows past the image: zeros (phase 2 stores nothing for them)
          if (!rok || j0 + 0 >= N) o0 = 0.f;
          if (!rok || j0 + 1 >= N) o1 = 0.f;
          if (!rok || j0 + 2 >= N) o2 = 0.f;
          if (!rok || j0 + 3 >= N) o3 = 0.f;
          *reinterpret_cast<uint2*>(prow + j0) = make_uint2(af_pk2(o0, o1), af_pk2(o2, o3));
          if (nt & 1) __builtin_amdgcn_sched_barrier(0);      // (the tiles' hash / tag / store chains scheduled together need > 256 registers)
        }
      }
    }
    __syncthreads();
    // ================= phase 2: wave = 4 rows, all heads =================
    {
      // the mix coefficients in VECTOR registers, fetched HERE for every item: through an address the compiler cannot prove uniform
      // (as scalar loads they are the pattern of DESIGN 2a) and cannot hoist out of the item loop (72 registers live across phase 1)
      int vz;
      asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
      float Wk[H][H], cin[H];
#pragma unroll
      for (int gg = 0; gg < H; ++gg) {
        float sacc = 0.f;
#pragma unroll
        for (int hh = 0; hh < H; ++hh) { const float w = a.W[gg * H + hh + vz]; Wk[gg][hh] = w * a.inv_keep; sacc += w; }
        cin[gg] = -sacc / (float)N;      // -shift_g (mix_stats_mm_kernel)
      }
      const unsigned hs = (unsigned)N * (unsigned)ld;
#pragma unroll 1
      for (int rr = 4 * wave + g; rr < QR; rr += 32) {       // row of the tile (16-row tiles: waves 4 - 7 have none)
        const int i = i0 + rr;
        const bool rok = i < N;
        const long long rbase = ((long long)b * H * N + min(i, N - 1)) * (long long)ld;
#pragma unroll 1
        for (int kg = l15; 4 * kg < ld; kg += 16) {            // key quads of the row (the quads past N inside ld are zeros)
          uint2 p[H];
#pragma unroll
          for (int hh = 0; hh < H; ++hh) p[hh] = *reinterpret_cast<const uint2*>(PL + ((long long)hh * QR + rr) * LP + 4 * kg);
          if (rok) {
#pragma unroll
            for (int hh = 0; hh < H; ++hh) *reinterpret_cast<uint2*>(a.Ps + rbase + (long long)hh * hs + 4 * kg) = p[hh];
          }
          float pv[H][4];
#pragma unroll
          for (int hh = 0; hh < H; ++hh) {
            const unsigned x = af_keep_pos(p[hh].x), y = af_keep_pos(p[hh].y);
            pv[hh][0] = af_lo(x); pv[hh][1] = af_hi(x); pv[hh][2] = af_lo(y); pv[hh][3] = af_hi(y);
          }
#pragma unroll
          for (int gg = 0; gg < H; ++gg) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = cin[gg];
#pragma unroll
              for (int hh = 0; hh < H; ++hh) v = fmaf(Wk[gg][hh], pv[hh][e], v);
              v = (rok && 4 * kg + e < N) ? v : 0.f;
              o[e] = v;
              s1[gg] += v; s2[gg] = fmaf(v, v, s2[gg]);
            }
            if (rok) *reinterpret_cast<uint2*>(a.Ac + rbase + (long long)gg * hs + 4 * kg) = make_uint2(af_pk2(o[0], o[1]), af_pk2(o[2], o[3]));
          }
        }
      }
    }
    __syncthreads();          // the tile is rewritten by the next item
  }
  // ---- the workgroup's shifted moments ----
#pragma unroll
  for (int gg = 0; gg < H; ++gg) {
    float v1 = s1[gg], v2 = s2[gg];
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) { v1 += __shfl_xor(v1, m, 64); v2 += __shfl_xor(v2, m, 64); }
    if (lane == 0) { red[wave][gg] = v1; red[wave][H + gg] = v2; }
  }
  __syncthreads();
  if (tid < 16) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[w][tid];
    a.partials[blockIdx.x * 16 + tid] = v;
  }
}

inline bool f1_on() { static const bool v = [] { const char* e = getenv("VU_ATTN_F1"); return !(e && e[0] == '0'); }(); return v; }

template <int NT, int KS, int QT>
int f1_launch(f1_args& a, int* nblocks, hipStream_t st) {
  constexpr int QR = 16 * QT, LP = NT * 16 + 8;
  const size_t lds = (size_t)8 * QR * LP * 2;
  auto kern = attn_f1_kernel<NT, KS, QT>;
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    vu_set_error("attn_f1: cannot reserve %zu bytes of LDS", lds);
    return VU_ELAUNCH;
  }
  a.nqt = (a.N + QR - 1) / QR;
  a.nitems = a.B * a.nqt;
  int grid = a.nitems;
  if (grid > 1024) grid = 1024 / 8 * 8;          // (the partials buffer holds 1024 workgroups; workgroups then walk several items)
  *nblocks = grid;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
  return VU_OK;
}

}  // namespace

// Covered: bf16, 8 heads, (N <= 208, d = 96) and (N <= 64, d = 384).  The probabilities and the centred map land in Ps / Ac exactly as
// vu_k_attn_scores + vu_k_mix_stats_mm(Ac) would leave them; `partials` gets *nblocks rows of 16 moments for vu_k_bn_finalize.
bool vu_attn_f1_ok(int dtype, int B, int N, int D, int H, int ld) {
  if (!f1_on() || dtype != 1 || H != 8 || ld % 8 != 0 || ld < N) return false;
  if ((double)B * H * N * (double)ld >= 4294967295.0) return false;          // 32-bit mask index
  const int d = D / H;
  // Level 1 (d = 96, N = 196) is instantiated and correct but NOT taken by default: measured 69 us against 35 + 20 for the two
  // kernels it replaces - at this size the work is the element-wise chain (scale, round, exp2, two 32-bit hash multiplies per key
  // pair, tag, 64 multiply-adds of the mix per position: ~45 VALU instructions per map element, a floor of ~26 us for the 19.7 M
  // elements of a module at 64 images), fusing removes map passes that were not the bound, and 448 workgroups of 8 waves on 256 CUs
  // balance worse than the two grids they replace.  VU_ATTN_F1=2 takes it for the record.  Level 0 (d = 384, N = 49): 20 us against
  // 18 + 8.
  static const bool lvl1 = [] { const char* e = getenv("VU_ATTN_F1"); return e && e[0] == '2'; }();
  return (lvl1 && d == 96 && N > 64 && N <= 208 && ld <= 208) || (d == 384 && N >= 32 && N <= 64 && ld <= 64);
}

int vu_k_attn_f1(const void* q, const void* k, void* Ps, void* Ac, const float* W, float* partials, int* nblocks, int B, int N, int D, int ld,
                 float scale, vu_rng rng, hipStream_t st) {
  f1_args a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.Ps = (bf16_t*)Ps; a.Ac = (bf16_t*)Ac; a.W = W; a.partials = partials;
  a.B = B; a.N = N; a.D = D; a.ld = ld; a.scale = scale; a.inv_keep = rng.inv_keep; a.rng = rng;
  const int d = D / 8;
  if (d == 96) return f1_launch<13, 3, 2>(a, nblocks, st);
  return f1_launch<4, 12, 1>(a, nblocks, st);
}
