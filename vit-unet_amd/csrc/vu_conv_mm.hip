// K4 on the matrix cores (round 3): the q / k / v convolutions of a block and the sum of their data gradients, bf16 storage,
// C in {1, 3}, patch sizes 8 / 16 / 32 (model.py:137-139,152-154: 3x3, zero halo at the PATCH border, no bias).
//
// The VALU form (vu_conv.hip) spends 243 multiply-adds per pixel on q, k and v (27 taps x 9 outputs) and is bound by that, not
// by its bytes (28 us for 77 MB at Base level 2).  Here a wave stages whole patches into LDS with a zero halo ([C][S+2][S+8]
// bf16, the data at column 4 so that a 16-byte row segment lands 8-byte aligned) and 16 pixels become the 16 rows of ONE
// v_mfma_f32_16x16x32_bf16: the k axis is the (channel, dy, dx) tap index (27 of 32 used), gathered from the halo image with
// eight 2-byte LDS reads per lane; the 16 columns are the 3 C outputs of q, k and v together (9 of 16 used).  The fp32 weights
// enter as a bf16 hi + lo pair (two MFMAs), so the products carry 16 weight bits like the fp32 FMAs did.  Cross attention
// (q from one tensor, k / v from another) is a second gather into the same accumulator (disjoint columns).
// The data gradient is the same gather over the three staged gradients with flipped taps: dX[ci] = sum_t sum_co,a,b
// w_t[co][ci][2-a][2-b] dOut_t[co][y+a-1][x+b-1], three (hi, lo) MFMA pairs into one accumulator whose columns are the input
// channels (cross form: dxq in columns 0..C-1, dxkv in C..2C-1).
#include <stdlib.h>
#include "vu_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
// (the halo image is written as 8-byte words and read back as 2-byte elements by the SAME wave: LDS keeps a wave's accesses in
// order, the compiler must too)
#define LDS_FENCE() asm volatile("" ::: "memory")

template <int C, int S, int DIV> struct CM {
  static constexpr int E = C * S * S;                      // elements of a patch
  static constexpr int U = E / 8;                          // 16-byte units of a patch
  static constexpr int PP0 = (192 / U) > 0 ? 192 / U : 1;
  static constexpr int PP = (PP0 / DIV) > 0 ? PP0 / DIV : 1;   // patches per wave trip
  static constexpr int NL = (PP * U + 63) / 64;            // 16-byte loads per lane and trip
  static constexpr int PITCH = S + 8, ROWS = S + 2, PLANE = ROWS * PITCH, PATCH = C * PLANE;
  static constexpr int WAVE = PP * PATCH;                  // LDS elements per wave and staged tensor
  static constexpr int UPR = S / 8;                        // units per plane row
  static constexpr int PXR = S < 16 ? S : 16;              // pixels of a 16-pixel group that share a row
  static constexpr int RG = 16 / PXR, GR = S / PXR;        // rows per group, groups per row
  static constexpr int GROUPS = S * S / 16;
};

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, b2));
}

// the lane's staging slots of a trip: unit u = lane + 64 i  ->  LDS element offset (or -1)
template <typename M, int C, int S>
__device__ __forceinline__ void staging_offsets(int lane, int (&off)[M::NL]) {
#pragma unroll
  for (int i = 0; i < M::NL; ++i) {
    const int u = lane + 64 * i;
    const int p = u / M::U, r = u % M::U;
    const int c = r / (S * M::UPR), r2 = r % (S * M::UPR);
    const int y = r2 / M::UPR, xu = r2 % M::UPR;
    off[i] = u < M::PP * M::U ? p * M::PATCH + c * M::PLANE + (y + 1) * M::PITCH + 4 + 8 * xu : -1;
  }
}
template <typename M>
__device__ __forceinline__ void fetch(const bf16_t* __restrict__ in, long long blk, int npatch, int lane, const int (&off)[M::NL],
                                      u32x4_t (&v)[M::NL]) {
#pragma unroll
  for (int i = 0; i < M::NL; ++i) {
    const int u = lane + 64 * i;
    const bool ok = off[i] >= 0 && blk * M::PP + u / M::U < npatch;
    v[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (ok) v[i] = *reinterpret_cast<const u32x4_t*>(in + (blk * M::PP * (long long)M::E + (long long)u * 8));
  }
}
template <typename M>
__device__ __forceinline__ void commit(unsigned short* L, const int (&off)[M::NL], const u32x4_t (&v)[M::NL]) {
#pragma unroll
  for (int i = 0; i < M::NL; ++i)
    if (off[i] >= 0) {
      *reinterpret_cast<u32x2_t*>(L + off[i]) = u32x2_t{v[i][0], v[i][1]};
      *reinterpret_cast<u32x2_t*>(L + off[i] + 4) = u32x2_t{v[i][2], v[i][3]};
    }
}
__device__ __forceinline__ bf16x8 gather8(const unsigned short* base, const int (&koff)[8]) {
  u32x4_t w;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = (unsigned)base[koff[2 * j]] | ((unsigned)base[koff[2 * j + 1]] << 16);
  return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ f32x4 mfma32(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// tap offsets of the lane's k block (k = 8 g4 + j = c 9 + a 3 + b) relative to the pixel's own position in the halo image
template <typename M, int C>
__device__ __forceinline__ void tap_offsets(int g4, int (&koff)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g4 + j;
    const int kk = k < 9 * C ? k : 4;                      // (unused k: any finite in-range element, its weight is 0)
    const int c = kk / 9, r = kk % 9;
    koff[j] = c * M::PLANE + (r / 3) * M::PITCH + (r % 3) + 3;
  }
}
// hi / lo bf16 halves of the lane's 8 weights of column n: w[k] for k = 8 g4 .. 8 g4 + 7 (nullptr / k >= 9 C: zeros)
template <int C, bool FLIP>
__device__ __forceinline__ void weight_op(const float* __restrict__ w, int col, int g4, bf16x8& hi, bf16x8& lo) {
  // forward: w[co = col][k]                       -> w + col * 9 C + k
  // data gradient (FLIP): k = co 9 + a 3 + b, w[co][ci = col][2 - a][2 - b] -> w + (co C + col) 9 + 8 - (a 3 + b)
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g4 + j;
    float x = 0.f;
    if (w && k < 9 * C) x = FLIP ? w[((k / 9) * C + col) * 9 + 8 - (k % 9)] : w[col * 9 * C + k];
    v[j] = x;
  }
  u32x4_t h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pack2(v[2 * j], v[2 * j + 1]);
    const float r0 = v[2 * j] - __uint_as_float(h[j] << 16), r1 = v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u);
    l[j] = pack2(r0, r1);
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

// Results leave through a wave-private LDS tile so that a store instruction writes whole 128-byte lines: a 16-pixel group is
// 32 bytes of a channel plane, four consecutive groups are one line; written straight from the accumulators (36 lanes x 8
// bytes per group) the outputs reached the L2 as quarter-line requests and the kernel ran at the VALU form's speed.
// A flush block = 8 groups = 2 line sets (S = 8: two patches; S >= 16: 128 pixels of one patch) x NC columns.
template <int C, int S, int NC> struct OutStage {
  static constexpr int LP = 72;                            // line pitch (elements): 16-byte aligned, 4 banks apart
  static constexpr int LINES = 2 * NC, UNITS = LINES * 8, NU = (UNITS + 63) / 64, ELEMS = LINES * LP;
};
// element offset of group gg (0..7) of a flush block relative to the block's first group, in the halo image
template <typename M, int S>
__device__ __forceinline__ constexpr int group_off(int gg) {
  return S == 8 ? (gg / 4) * M::PATCH + (gg % 4) * 2 * M::PITCH : (gg / M::GR) * M::RG * M::PITCH + (gg % M::GR) * M::PXR;
}

// ---- forward: q = conv(in0, wq), k = conv(in1, wk), v = conv(in1, wv) -------------------------------------------------
template <int C, int S, int NIN>
__global__ __launch_bounds__(256) void conv_qkv_mm_kernel(const bf16_t* __restrict__ in0, const bf16_t* __restrict__ in1,
                                                          const float* __restrict__ wq, const float* __restrict__ wk,
                                                          const float* __restrict__ wv, bf16_t* __restrict__ oq,
                                                          bf16_t* __restrict__ ok, bf16_t* __restrict__ ov, int npatch) {
  typedef CM<C, S, NIN> M;
  typedef OutStage<C, S, 3 * C> OS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g4 = lane >> 4;
  unsigned short* L0 = reinterpret_cast<unsigned short*>(smem_raw) + wave * (NIN * M::WAVE + OS::ELEMS);
  unsigned short* L1 = L0 + (NIN - 1) * M::WAVE;
  unsigned short* Lo = L0 + NIN * M::WAVE;
  for (int i = lane; i < NIN * M::WAVE / 8; i += 64) reinterpret_cast<u32x4_t*>(L0)[i] = u32x4_t{0u, 0u, 0u, 0u};
  int soff[M::NL], koff[8];
  staging_offsets<M, C, S>(lane, soff);
  tap_offsets<M, C>(g4, koff);
  // weight operands: column n = t C + co.  One input: all nine columns in one operand; cross: q | (k, v) separately
  const int t = l15 / C, co = l15 % C;
  const float* wsel = t == 0 ? wq : (t == 1 ? wk : (t == 2 ? wv : nullptr));
  bf16x8 bh0, bl0, bh1, bl1;
  weight_op<C, false>(NIN == 1 ? wsel : (t == 0 ? wq : nullptr), co, g4, bh0, bl0);
  if (NIN == 2) weight_op<C, false>(t == 1 ? wk : (t == 2 ? wv : nullptr), co, g4, bh1, bl1);
  const bool ocol = l15 < 3 * C;
  const int pixoff = (l15 / M::PXR) * M::PITCH + (l15 % M::PXR);
  const int lo_w = l15 * OS::LP + 4 * g4;                   // the lane's slot in line (set 0, column l15)
  // flush slots of the lane: unit u = lane + 64 i -> line u / 8 = set NC + column, 16 bytes at u % 8
  bf16_t* fptr[OS::NU];
  int flds[OS::NU], fset[OS::NU];
#pragma unroll
  for (int i = 0; i < OS::NU; ++i) {
    const int u = lane + 64 * i, line = u / 8, set = line / (3 * C), tc = line % (3 * C);
    bf16_t* ob = tc / C == 0 ? oq : (tc / C == 1 ? ok : ov);
    fptr[i] = ob + (tc % C) * (S * S) + (u % 8) * 8 + (S == 8 ? 0 : 64 * set);
    flds[i] = line * OS::LP + (u % 8) * 8;
    fset[i] = u < OS::UNITS ? set : -1;
  }
  const long long nblk = (npatch + M::PP - 1) / M::PP;
  u32x4_t v0[M::NL], v1[M::NL];
  long long blk = (long long)blockIdx.x * 4 + wave;
  const long long stride = (long long)gridDim.x * 4;
  if (blk < nblk) {
    fetch<M>(in0, blk, npatch, lane, soff, v0);
    if (NIN == 2) fetch<M>(in1, blk, npatch, lane, soff, v1);
  }
  for (; blk < nblk; blk += stride) {
    commit<M>(L0, soff, v0);
    if (NIN == 2) commit<M>(L1, soff, v1);
    LDS_FENCE();
    if (blk + stride < nblk) {                               // next trip in flight during the products
      fetch<M>(in0, blk + stride, npatch, lane, soff, v0);
      if (NIN == 2) fetch<M>(in1, blk + stride, npatch, lane, soff, v1);
    }
    constexpr int NB = M::PP * M::GROUPS / 8;
#pragma unroll 1
    for (int fb = 0; fb < NB; ++fb) {
      const int p0 = (fb * 8) / M::GROUPS, g0 = (fb * 8) % M::GROUPS;
      const long long patch0 = blk * M::PP + p0;
      if (patch0 >= npatch) break;
      const int base = p0 * M::PATCH + (S == 8 ? 0 : (g0 / M::GR) * M::RG * M::PITCH) + pixoff;
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) {
        const int goff = base + group_off<M, S>(gg);
        const bf16x8 a0 = gather8(L0 + goff, koff);
        f32x4 acc = mfma32(a0, bh0, f32x4{0.f, 0.f, 0.f, 0.f});
        acc = mfma32(a0, bl0, acc);
        if (NIN == 2) {
          const bf16x8 a1 = gather8(L1 + goff, koff);
          acc = mfma32(a1, bh1, acc);
          acc = mfma32(a1, bl1, acc);
        }
        if (ocol)
          *reinterpret_cast<u32x2_t*>(Lo + lo_w + (gg / 4) * (3 * C) * OS::LP + (gg % 4) * 16) = u32x2_t{pack2(acc[0], acc[1]), pack2(acc[2], acc[3])};
      }
      LDS_FENCE();
      const long long gbase = patch0 * M::E + (S == 8 ? 0 : g0 * 16);
#pragma unroll
      for (int i = 0; i < OS::NU; ++i)
        if (fset[i] >= 0 && (S != 8 || patch0 + fset[i] < npatch))
          *reinterpret_cast<u32x4_t*>(fptr[i] + gbase + (S == 8 ? fset[i] * M::E : 0)) = *reinterpret_cast<const u32x4_t*>(Lo + flds[i]);
      LDS_FENCE();
    }
  }
}

// ---- data gradient: dxq (+ dxkv) from dq, dk, dv ------------------------------------------------------------------------
// CROSS = false: din0 = convT(dq, wq) + convT(dk, wk) + convT(dv, wv) + add0
// CROSS = true : din0 = convT(dq, wq) + add0 ; din1 = convT(dk, wk) + convT(dv, wv) + add1
template <int C, int S, bool CROSS>
__global__ __launch_bounds__(256) void conv_qkv_dgrad_mm_kernel(const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk,
                                                                const bf16_t* __restrict__ dv, const float* __restrict__ wq,
                                                                const float* __restrict__ wk, const float* __restrict__ wv,
                                                                const bf16_t* __restrict__ add0, const bf16_t* __restrict__ add1,
                                                                bf16_t* __restrict__ din0, bf16_t* __restrict__ din1, int npatch) {
  typedef CM<C, S, 4> M;
  constexpr int NC = (CROSS ? 2 : 1) * C;
  typedef OutStage<C, S, NC> OS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g4 = lane >> 4;
  unsigned short* L0 = reinterpret_cast<unsigned short*>(smem_raw) + wave * (3 * M::WAVE + OS::ELEMS);
  unsigned short* L1 = L0 + M::WAVE;
  unsigned short* L2 = L1 + M::WAVE;
  unsigned short* Lo = L2 + M::WAVE;
  for (int i = lane; i < 3 * M::WAVE / 8; i += 64) reinterpret_cast<u32x4_t*>(L0)[i] = u32x4_t{0u, 0u, 0u, 0u};
  int soff[M::NL], koff[8];
  staging_offsets<M, C, S>(lane, soff);
  tap_offsets<M, C>(g4, koff);
  // columns: ci (all three gradients) or, CROSS, ci for dq and C + ci for dk / dv
  const int half = l15 / C, ci = l15 % C;
  bf16x8 bh[3], bl[3];
  weight_op<C, true>(half == 0 ? wq : nullptr, ci, g4, bh[0], bl[0]);
  weight_op<C, true>(half == (CROSS ? 1 : 0) ? wk : nullptr, ci, g4, bh[1], bl[1]);
  weight_op<C, true>(half == (CROSS ? 1 : 0) ? wv : nullptr, ci, g4, bh[2], bl[2]);
  const bool ocol = l15 < NC;
  const bf16_t* abase = half == 0 ? add0 : add1;
  const int pixoff = (l15 / M::PXR) * M::PITCH + (l15 % M::PXR);
  const int lo_w = l15 * OS::LP + 4 * g4;
  bf16_t* fptr[OS::NU];
  int flds[OS::NU], fset[OS::NU];
#pragma unroll
  for (int i = 0; i < OS::NU; ++i) {
    const int u = lane + 64 * i, line = u / 8, set = line / NC, tc = line % NC;
    bf16_t* ob = tc / C == 0 ? din0 : din1;
    fptr[i] = ob + (tc % C) * (S * S) + (u % 8) * 8 + (S == 8 ? 0 : 64 * set);
    flds[i] = line * OS::LP + (u % 8) * 8;
    fset[i] = u < OS::UNITS ? set : -1;
  }
  const long long nblk = (npatch + M::PP - 1) / M::PP;
  u32x4_t v0[M::NL], v1[M::NL], v2[M::NL];
  long long blk = (long long)blockIdx.x * 4 + wave;
  const long long stride = (long long)gridDim.x * 4;
  if (blk < nblk) {
    fetch<M>(dq, blk, npatch, lane, soff, v0);
    fetch<M>(dk, blk, npatch, lane, soff, v1);
    fetch<M>(dv, blk, npatch, lane, soff, v2);
  }
  for (; blk < nblk; blk += stride) {
    commit<M>(L0, soff, v0);
    commit<M>(L1, soff, v1);
    commit<M>(L2, soff, v2);
    LDS_FENCE();
    if (blk + stride < nblk) {
      fetch<M>(dq, blk + stride, npatch, lane, soff, v0);
      fetch<M>(dk, blk + stride, npatch, lane, soff, v1);
      fetch<M>(dv, blk + stride, npatch, lane, soff, v2);
    }
    constexpr int NB = M::PP * M::GROUPS / 8;
#pragma unroll 1
    for (int fb = 0; fb < NB; ++fb) {
      const int p0 = (fb * 8) / M::GROUPS, g0 = (fb * 8) % M::GROUPS;
      const long long patch0 = blk * M::PP + p0;
      if (patch0 >= npatch) break;
      const int base = p0 * M::PATCH + (S == 8 ? 0 : (g0 / M::GR) * M::RG * M::PITCH) + pixoff;
      const long long gbase = patch0 * M::E + (S == 8 ? 0 : g0 * 16);
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) {
        const int goff = base + group_off<M, S>(gg);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool pv = S != 8 || patch0 + gg / 4 < npatch;
        if (ocol && abase && pv) {
          const u32x2_t a = *reinterpret_cast<const u32x2_t*>(abase + gbase + ci * (S * S) + (S == 8 ? (gg / 4) * M::E + (gg % 4) * 16 : gg * 16) + 4 * g4);
          acc = f32x4{__uint_as_float(a[0] << 16), __uint_as_float(a[0] & 0xffff0000u), __uint_as_float(a[1] << 16), __uint_as_float(a[1] & 0xffff0000u)};
        }
        const bf16x8 a0 = gather8(L0 + goff, koff);
        acc = mfma32(a0, bh[0], acc);
        acc = mfma32(a0, bl[0], acc);
        const bf16x8 a1 = gather8(L1 + goff, koff);
        acc = mfma32(a1, bh[1], acc);
        acc = mfma32(a1, bl[1], acc);
        const bf16x8 a2 = gather8(L2 + goff, koff);
        acc = mfma32(a2, bh[2], acc);
        acc = mfma32(a2, bl[2], acc);
        if (ocol) *reinterpret_cast<u32x2_t*>(Lo + lo_w + (gg / 4) * NC * OS::LP + (gg % 4) * 16) = u32x2_t{pack2(acc[0], acc[1]), pack2(acc[2], acc[3])};
      }
      LDS_FENCE();
#pragma unroll
      for (int i = 0; i < OS::NU; ++i)
        if (fset[i] >= 0 && (S != 8 || patch0 + fset[i] < npatch))
          *reinterpret_cast<u32x4_t*>(fptr[i] + gbase + (S == 8 ? fset[i] * M::E : 0)) = *reinterpret_cast<const u32x4_t*>(Lo + flds[i]);
      LDS_FENCE();
    }
  }
}

// NOT TAKEN by default (VU_CONV_MM=1 enables it wherever instantiated, for measurements): the tap gather is bound by its
// 2-byte LDS reads (8 per MFMA, ~6 cycles each with the bank conflicts of four tap groups), so back to back on 64 images the
// forward takes 28.4 / 24.3 / 25.4 us at patch size 8 / 16 / 32 against 29.0 / 29.1 / 28.2 for the stencil form and the data
// gradient (24 reads per group) 45 - 112 us against 35; on the Base step the forward at s >= 16 was worth +0.3 %, inside the
// noise, and its differently rounded q / k moved the ill-conditioned first level-1 decoder block ("saturated rows")
// from 0.05 to 0.09 in the teacher-forced dx check.  Kept as the measured answer to "convolutions on the matrix cores".
inline int mm_mode() { static const int v = [] { const char* e = getenv("VU_CONV_MM"); return (e && e[0] == '1') ? 1 : 0; }(); return v; }

template <typename K>
inline int reserve(K kern, size_t lds) {
  if (lds > 65536 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    vu_set_error("conv3x3 (MFMA form): cannot reserve %zu bytes of LDS", lds);
    return VU_ELAUNCH;
  }
  return VU_OK;
}

template <int C, int S>
int launch_fwd(const bf16_t* xq, const bf16_t* xkv, const float* wq, const float* wk, const float* wv, bf16_t* q, bf16_t* k, bf16_t* v,
               long long npatch, hipStream_t st) {
  const bool same = xq == xkv;
  if (same) {
    typedef CM<C, S, 1> M;
    const long long nblk = (npatch + M::PP - 1) / M::PP;
    const int grid = (int)((nblk + 7) / 8 < 1 ? 1 : ((nblk + 7) / 8 > 2048 ? 2048 : (nblk + 7) / 8));      // ~2 trips per wave
    const size_t lds = (size_t)4 * (M::WAVE + OutStage<C, S, 3 * C>::ELEMS) * 2;
    auto kern = conv_qkv_mm_kernel<C, S, 1>;
    if (int e = reserve(kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, xq, xkv, wq, wk, wv, q, k, v, (int)npatch);
  } else {
    typedef CM<C, S, 2> M;
    const long long nblk = (npatch + M::PP - 1) / M::PP;
    const int grid = (int)((nblk + 7) / 8 < 1 ? 1 : ((nblk + 7) / 8 > 2048 ? 2048 : (nblk + 7) / 8));
    const size_t lds = (size_t)4 * (2 * M::WAVE + OutStage<C, S, 3 * C>::ELEMS) * 2;
    auto kern = conv_qkv_mm_kernel<C, S, 2>;
    if (int e = reserve(kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, xq, xkv, wq, wk, wv, q, k, v, (int)npatch);
  }
  return VU_OK;
}

template <int C, int S>
int launch_dgrad(const bf16_t* dq, const bf16_t* dk, const bf16_t* dv, const float* wq, const float* wk, const float* wv,
                 const bf16_t* add_q, const bf16_t* add_kv, bf16_t* dxq, bf16_t* dxkv, long long npatch, hipStream_t st) {
  typedef CM<C, S, 4> M;
  const long long nblk = (npatch + M::PP - 1) / M::PP;
  const int grid = (int)((nblk + 7) / 8 < 1 ? 1 : ((nblk + 7) / 8 > 2048 ? 2048 : (nblk + 7) / 8));
  const size_t lds = (size_t)4 * (3 * M::WAVE + OutStage<C, S, 2 * C>::ELEMS) * 2;
  if (dxkv) {
    auto kern = conv_qkv_dgrad_mm_kernel<C, S, true>;
    if (int e = reserve(kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, dq, dk, dv, wq, wk, wv, add_q, add_kv, dxq, dxkv, (int)npatch);
  } else {
    auto kern = conv_qkv_dgrad_mm_kernel<C, S, false>;
    if (int e = reserve(kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, dq, dk, dv, wq, wk, wv, add_q, add_kv, dxq, dxkv, (int)npatch);
  }
  return VU_OK;
}

}  // namespace

// covered: bf16 storage, C in {1, 3}, patch size 8 / 16 / 32, 16-byte aligned tensors
bool vu_conv_mm_ok(int dtype, int C, int s, int backward) {
  const bool inst = dtype == 1 && (C == 1 || C == 3) && (s == 8 || s == 16 || s == 32);
  (void)backward;
  return inst && mm_mode() == 1;
}

#define VU_CONV_MM_DISPATCH(FN, ...)                                                          \
  do {                                                                                        \
    if (C == 3 && s == 8) return FN<3, 8>(__VA_ARGS__);                                       \
    if (C == 3 && s == 16) return FN<3, 16>(__VA_ARGS__);                                     \
    if (C == 3 && s == 32) return FN<3, 32>(__VA_ARGS__);                                     \
    if (C == 1 && s == 8) return FN<1, 8>(__VA_ARGS__);                                       \
    if (C == 1 && s == 16) return FN<1, 16>(__VA_ARGS__);                                     \
    if (C == 1 && s == 32) return FN<1, 32>(__VA_ARGS__);                                     \
    vu_set_error("conv3x3 (MFMA form): C=%d s=%d not instantiated", C, s);                    \
    return VU_EUNSUPPORTED;                                                                   \
  } while (0)

int vu_k_conv_mm_qkv_fwd(const void* xq, const void* xkv, const float* wq, const float* wk, const float* wv, void* q, void* k, void* v,
                         long long npatch, int C, int s, hipStream_t st) {
  VU_CONV_MM_DISPATCH(launch_fwd, (const bf16_t*)xq, (const bf16_t*)xkv, wq, wk, wv, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, npatch, st);
}
int vu_k_conv_mm_qkv_dgrad(const void* dq, const void* dk, const void* dv, const float* wq, const float* wk, const float* wv,
                           const void* add_q, const void* add_kv, void* dxq, void* dxkv, long long npatch, int C, int s, hipStream_t st) {
  VU_CONV_MM_DISPATCH(launch_dgrad, (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)dv, wq, wk, wv, (const bf16_t*)add_q,
                      (const bf16_t*)add_kv, (bf16_t*)dxq, (bf16_t*)dxkv, npatch, st);
}
