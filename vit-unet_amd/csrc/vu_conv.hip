// K4 / K15: 3x3 convolutions with zero halo at the PATCH border (model.py:137-139,152-154 per
// patch; :370,:428 whole image = one patch of size im).  HBM-bound stencils (K = 9C <= 27 taps,
// C <= 4 outputs: not GEMM-shaped).  One thread owns 4 consecutive pixels of a patch row and
// loads each channel's 3x6 window with three unconditional loads per row (one 4-vector + two
// clamped halo scalars) so all loads of a thread are in flight together; q, k and v are produced
// from one read of x; their data gradients are summed in one kernel.
#include <stdlib.h>
#include <string.h>
#include "vu_kernels.h"

namespace {

// window of rows y-1..y+1, columns x0-1..x0+4 of one s x s channel plane; zero outside.
template <typename T>
__device__ __forceinline__ void load_win(const T* __restrict__ plane, int s, int y, int x0, float (&w)[3][6]) {
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = y + dy - 1;
    const bool rv = (yy >= 0) && (yy < s);
    const int yc = yy < 0 ? 0 : (yy >= s ? s - 1 : yy);
    const T* rp = plane + yc * s + x0;
    const vu_f4 c = vu_ld4(rp);
    const bool lv = x0 > 0, rr = x0 + 4 < s;
    const float l = vu_ld(rp - (lv ? 1 : 0));
    const float r = vu_ld(rp + (rr ? 4 : 3));
    w[dy][0] = (rv && lv) ? l : 0.f;
    w[dy][1] = rv ? c.v[0] : 0.f;
    w[dy][2] = rv ? c.v[1] : 0.f;
    w[dy][3] = rv ? c.v[2] : 0.f;
    w[dy][4] = rv ? c.v[3] : 0.f;
    w[dy][5] = (rv && rr) ? r : 0.f;
  }
}

// Neighbour-lane forms of the window (consecutive lanes hold consecutive pixel quads of a patch, row-major, and a patch
// row of s/4 quads never straddles a 16-lane DPP row when s/4 divides 16):
//   SH = 1: the two halo pixels of every row come from lanes -1 / +1 (row_shr:1 / row_shl:1) instead of two scalar loads;
//   SH = 2: (s == 8, a DPP row is one whole 8x8 plane) rows y-1 / y+1 come from lanes -2 / +2 as well: one load per plane.
// The stencil is issue-bound on its many narrow loads (27 per thread for C = 3), not on bytes.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <typename T, int SH>
__device__ __forceinline__ void load_win_sh(const T* __restrict__ plane, int s, int y, int x0, float (&w)[3][6]) {
  if constexpr (SH == 0) { load_win(plane, s, y, x0, w); return; }
  if constexpr (SH == 2) {
    const vu_f4 c = vu_ld4(plane + y * s + x0);
    const bool up = y > 0, dn = y < s - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u = dpp_mov<0x112>(c.v[i]), d = dpp_mov<0x102>(c.v[i]);   // row_shr:2 / row_shl:2
      w[0][1 + i] = up ? u : 0.f;
      w[1][1 + i] = c.v[i];
      w[2][1 + i] = dn ? d : 0.f;
    }
  } else {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      const bool rv = (yy >= 0) && (yy < s);
      const int yc = yy < 0 ? 0 : (yy >= s ? s - 1 : yy);
      const vu_f4 c = vu_ld4(plane + yc * s + x0);
#pragma unroll
      for (int i = 0; i < 4; ++i) w[dy][1 + i] = rv ? c.v[i] : 0.f;
    }
  }
  const bool lv = x0 > 0, rr = x0 + 4 < s;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const float l = dpp_mov<0x111>(w[dy][4]), r = dpp_mov<0x101>(w[dy][1]);   // row_shr:1 / row_shl:1
    w[dy][0] = lv ? l : 0.f;
    w[dy][5] = rr ? r : 0.f;
  }
}
inline int conv_shuffle_form(int s) { return s == 8 ? 2 : ((s >= 8 && s <= 64 && 64 % s == 0) ? 1 : 0); }

__device__ __forceinline__ void quad_coords(long long qid, int s, long long& patch, int& y, int& x0) {
  // 32-bit unsigned arithmetic (the launchers guarantee nquads < 2^32): a 64-bit division by a
  // run-time value costs >100 instructions per quad
  const unsigned qpp = (unsigned)(s * s) >> 2;   // quads per channel plane
  const unsigned q = (unsigned)qid;
  const unsigned pch = q / qpp;
  const unsigned rem = (q - pch * qpp) << 2;
  const unsigned yy = rem / (unsigned)s;
  patch = pch;
  y = (int)yy;
  x0 = (int)(rem - yy * (unsigned)s);
}

// ---- forward ---------------------------------------------------------------------------------
// NOUT = 1: out0 = conv(in0, w0) (+bias)      NOUT = 3: q = conv(in0,w0), k = conv(in1,w1), v = conv(in1,w2)
// WL: where the weights of the rolled (o, co) loops come from.  0: scalar loads off the kernel-argument pointers inside the
// loops (rounds 1 - 3).  1: all NOUT * 9 C C weights staged ONCE per workgroup into LDS by vector loads, each (o, co) trip
// reads its 9 C weights back as broadcast ds_read_b128s (every lane the same address).  Round 4 found the scalar form
// returning WRONG results for whole waves (one pixel of every quad of the wave) in 4 - 10 % of the launches whenever a SECOND
// PROCESS was computing on the same GPU (tools/contention_ops.py; the idle-GPU suites never saw it) - the cause of the red
// two-rank rehearsal of round 3.  The LDS form is the default; VU_CONV_W=smem selects the scalar form for the A/B record.
template <typename TI, typename TO, int C, int NOUT, int SH = 0, int WL = 0>
#ifndef VU_CONV_WL_WAVES
#define VU_CONV_WL_WAVES 3
#endif
// (waves per SIMD the shuffle form is held to: 4 with scalar-load weights; the LDS-weight form spills 19 registers at 4 - VU_CONV_WL_WAVES)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((SH == 1 && NOUT == 3) ? (WL ? VU_CONV_WL_WAVES : 4) : 1, 8))) void conv_fwd_kernel(const TI* __restrict__ in0, const TI* __restrict__ in1,
                                                       const float* __restrict__ w0, const float* __restrict__ w1,
                                                       const float* __restrict__ w2, const float* __restrict__ bias,
                                                       TO* __restrict__ o0, TO* __restrict__ o1, TO* __restrict__ o2,
                                                       long long nquads, int s) {
  // WL = 0: weights are wave-uniform: indexed with compile-time constants off a kernel-argument pointer
  // they come through scalar loads into SGPRs (no LDS, no VGPRs)
  constexpr int NWC = 9 * C * C, NWP = (9 * C + 3) / 4 * 4;       // weights per convolution; per output channel, padded to 16 B
  __shared__ __attribute__((aligned(16))) float wl[WL ? NOUT * C * NWP : 4];
  __shared__ float bl[WL ? 4 : 1];
  if constexpr (WL != 0) {
    for (int i = threadIdx.x; i < NOUT * NWC; i += 256) {
      const int o = i / NWC, r = i - o * NWC, co = r / (9 * C), t = r - co * (9 * C);
      const float* wp = o == 0 ? w0 : (o == 1 ? w1 : w2);
      wl[(o * C + co) * NWP + t] = wp[r];
    }
    if (threadIdx.x < 4) bl[threadIdx.x] = (bias && threadIdx.x < C) ? bias[threadIdx.x] : 0.f;
    __syncthreads();
  }
  const int ss = s * s;
  const bool same = (in0 == in1);
  for (long long qid = blockIdx.x * (long long)blockDim.x + threadIdx.x; qid < nquads;
       qid += (long long)gridDim.x * blockDim.x) {
    long long patch; int y, x0;
    quad_coords(qid, s, patch, y, x0);
    const long long pbase = patch * (long long)(C * ss);
    const long long obase = pbase + y * s + x0;
    float win[C][3][6];
#pragma unroll
    for (int ci = 0; ci < C; ++ci) load_win_sh<TI, SH>(in0 + pbase + ci * ss, s, y, x0, win[ci]);
#pragma unroll 1
    for (int o = 0; o < NOUT; ++o) {
      if (NOUT == 3 && o == 1 && !same) {
#pragma unroll
        for (int ci = 0; ci < C; ++ci) load_win_sh<TI, SH>(in1 + pbase + ci * ss, s, y, x0, win[ci]);
      }
      TO* op = o == 0 ? o0 : (o == 1 ? o1 : o2);
      const float* __restrict__ wp = o == 0 ? w0 : (o == 1 ? w1 : w2);
#pragma unroll 1
      for (int co = 0; co < C; ++co) {     // (rolled: 27 weights live at a time; unrolled, the 81 of a convolution sit in VGPRs)
        vu_f4 acc;
        float wr[NWP];
        float b0;
        if constexpr (WL != 0) {
          const f32x4* wv4 = reinterpret_cast<const f32x4*>(wl + (o * C + co) * NWP);
#pragma unroll
          for (int t4 = 0; t4 < NWP / 4; ++t4) {
            const f32x4 x4 = wv4[t4];
            wr[4 * t4] = x4[0]; wr[4 * t4 + 1] = x4[1]; wr[4 * t4 + 2] = x4[2]; wr[4 * t4 + 3] = x4[3];
          }
          b0 = bl[co];
        } else {
#pragma unroll
          for (int t = 0; t < 9 * C; ++t) wr[t] = wp[co * C * 9 + t];
          b0 = bias ? bias[co] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc.v[i] = b0;
#pragma unroll
        for (int ci = 0; ci < C; ++ci)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float wv = wr[ci * 9 + ky * 3 + kx];
#pragma unroll
              for (int i = 0; i < 4; ++i) acc.v[i] += wv * win[ci][ky][i + kx];
            }
        vu_st4(op + obase + co * ss, acc);
      }
    }
  }
}

// ---- data gradient -----------------------------------------------------------------------------
// dIn[ci,y,x] = sum_co,ky,kx w[co][ci][ky][kx] * dOut[co, y-ky+1, x-kx+1]
// NIN = 1: din0 = convT(d0,w0) + add0
// NIN = 3: same input:  din0 = convT(d0,w0)+convT(d1,w1)+convT(d2,w2) + add0
//          cross:       din0 = convT(d0,w0) + add0 ; din1 = convT(d1,w1)+convT(d2,w2) + add1
template <typename TDO, typename T, int C, int NIN, int SH = 0>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const TDO* __restrict__ d0, const TDO* __restrict__ d1,
                                                         const TDO* __restrict__ d2, const float* __restrict__ w0,
                                                         const float* __restrict__ w1, const float* __restrict__ w2,
                                                         const T* add0, const T* add1, T* din0, T* din1,
                                                         long long nquads, int s) {
  const int ss = s * s;
  const bool cross = (NIN == 3) && (din1 != nullptr);
  for (long long qid = blockIdx.x * (long long)blockDim.x + threadIdx.x; qid < nquads;
       qid += (long long)gridDim.x * blockDim.x) {
    long long patch; int y, x0;
    quad_coords(qid, s, patch, y, x0);
    const long long pbase = patch * (long long)(C * ss);
    const long long obase = pbase + y * s + x0;
    float acc0[C][4], acc1[C][4];
#pragma unroll
    for (int ci = 0; ci < C; ++ci) {
      vu_f4 a0 = {{0.f, 0.f, 0.f, 0.f}}, a1 = {{0.f, 0.f, 0.f, 0.f}};
      if (add0) a0 = vu_ld4(add0 + obase + ci * ss);
      if (cross && add1) a1 = vu_ld4(add1 + obase + ci * ss);
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc0[ci][i] = a0.v[i]; acc1[ci][i] = a1.v[i]; }
    }
#pragma unroll 1
    for (int t = 0; t < NIN; ++t) {
      const TDO* dp = t == 0 ? d0 : (t == 1 ? d1 : d2);
      const float* __restrict__ wp = t == 0 ? w0 : (t == 1 ? w1 : w2);
      float win[C][3][6];
#pragma unroll
      for (int co = 0; co < C; ++co) load_win_sh<TDO, SH>(dp + pbase + co * ss, s, y, x0, win[co]);
      const bool to1 = cross && t > 0;
      float tmp[C][4];
#pragma unroll
      for (int ci = 0; ci < C; ++ci) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tmp[ci][i] = 0.f;
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float wv = wp[(co * C + ci) * 9 + ky * 3 + kx];
#pragma unroll
              for (int i = 0; i < 4; ++i) tmp[ci][i] += wv * win[co][2 - ky][i + 2 - kx];
            }
      }
#pragma unroll
      for (int ci = 0; ci < C; ++ci)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (to1) acc1[ci][i] += tmp[ci][i]; else acc0[ci][i] += tmp[ci][i];
        }
    }
#pragma unroll
    for (int ci = 0; ci < C; ++ci) {
      vu_f4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o.v[i] = acc0[ci][i];
      vu_st4(din0 + obase + ci * ss, o);
      if (cross) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[i] = acc1[ci][i];
        vu_st4(din1 + obase + ci * ss, o);
      }
    }
  }
}

// ---- weight gradient ---------------------------------------------------------------------------
// dW[co][ci][ky][kx] += sum_pix dOut[co,y,x] * in[ci,y+ky-1,x+kx-1] ; blockIdx.y selects the conv
struct WgradSet { const void* dout[3]; const void* in[3]; float* dw[3]; float* dbias[3]; };

template <typename TDO, typename TI, int C>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradSet set, long long nquads, int s, float* __restrict__ part) {
  // blockIdx.y = conv * C + co: one output channel of one convolution per block (27 accumulators
  // for C = 3 instead of 81: twice the waves per SIMD; the input windows come from L1/L2)
  constexpr int NW = C * 9;
  __shared__ float red[4][NW + 1];
  const int cv = blockIdx.y / C, co = blockIdx.y % C;
  const TDO* dout = (const TDO*)set.dout[cv];
  const TI* in = (const TI*)set.in[cv];
  float* dw = set.dw[cv] + co * NW;
  float* dbias = set.dbias[cv];
  const int ss = s * s;
  float acc[NW], accb = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) acc[i] = 0.f;
  for (long long qid = blockIdx.x * (long long)blockDim.x + threadIdx.x; qid < nquads;
       qid += (long long)gridDim.x * blockDim.x) {
    long long patch; int y, x0;
    quad_coords(qid, s, patch, y, x0);
    const long long pbase = patch * (long long)(C * ss);
    const long long obase = pbase + y * s + x0;
    float win[C][3][6];
#pragma unroll
    for (int ci = 0; ci < C; ++ci) load_win(in + pbase + ci * ss, s, y, x0, win[ci]);
    const vu_f4 d = vu_ld4(dout + obase + co * ss);
    accb += d.v[0] + d.v[1] + d.v[2] + d.v[3];
#pragma unroll
    for (int ci = 0; ci < C; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) a += d.v[i] * win[ci][ky][i + kx];
          acc[ci * 9 + ky * 3 + kx] += a;
        }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const float v = vu_wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = v;
  }
  {
    const float v = vu_wave_sum(accb);
    if (lane == 0) red[wave][NW] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NW + 1; i += blockDim.x) {
    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (part) part[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 32 + i] = v;     // summed in block order below
    else if (i < NW) atomicAdd(dw + i, v);
    else if (dbias) atomicAdd(dbias + co, v);
  }
}
// deterministic tail of conv_wgrad_kernel (C = 3: 27 weights + 1 bias sum per (conv, output channel)): one wave per output
// element adds the partials of the gridDim.x blocks in order.  (512 blocks per output channel spent ~25 us of the output
// convolution's 51 us in contended float atomics.)
template <int C>
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(WgradSet set, const float* __restrict__ part, int nblocks, int nconv) {
  constexpr int NW = C * 9;
  const int o = blockIdx.x * 16 + (threadIdx.x >> 6), lane = threadIdx.x & 63;       // o = (conv * C + co) * 32 + i
  const int y = o >> 5, i = o & 31;
  if (y >= nconv * C || i > NW) return;       // wave-uniform
  float s[4] = {0.f, 0.f, 0.f, 0.f};      // four chains over loads issued together (the order of the queued form, vu_tsgemm.hip)
  const float* src = part + (long long)y * nblocks * 32 + i;
  int b = lane;
  for (; b + 3 * 64 < nblocks; b += 4 * 64) {
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = src[(long long)(b + 64 * u) * 32];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += x[u];
  }
  for (; b < nblocks; b += 64) s[0] += src[(long long)b * 32];
  float a = (s[0] + s[1]) + (s[2] + s[3]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
  if (lane == 0) {
    const int cv = y / C, co = y % C;
    if (i < NW) set.dw[cv][co * NW + i] += a;
    else if (set.dbias[cv]) set.dbias[cv][co] += a;
  }
}

// ---------------------------------------------------------------------------------------------
// conv_wgrad_mm_kernel (bf16 storage, C = 3, q/k/v set): the weight gradients as a GEMM on the
// matrix cores,  dW_cv[co][tap] = sum_pixels dout_cv[co][pixel] * shifted_in[tap][pixel],
// tap = (ci, ky, kx), contraction over pixels.  One MFMA k-step is 32 pixels = 4 row segments of 8
// pixels (lane group lg = segment).  A operand: rows n = output channels, a lane's 8 k-slots are one
// 16-byte load of a dout row segment.  B operand: columns = taps; a lane loads the 8-pixel segment of
// input row y + ky - 1 of channel ci (plus the two edge pixels when the patch is wider than 8) and
// funnel-shifts it by kx - 1: the im2col matrix is never materialised.  q uses xq, k and v use xkv:
// two products (3 x 27 and 6 x 27), 2 tap tiles each.  Accumulators stay in registers over the
// block's pixel stream; one LDS reduction over the 4 waves and 9 x 27 float atomics per block.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 shift_segment(const uint4 v, unsigned left, unsigned right, int kx) {
  // kx = 0: [left, p0..p6] ; kx = 1: [p0..p7] ; kx = 2: [p1..p7, right]   (16-bit elements, p0 = low half of v.x)
  const uint4 a = make_uint4(__builtin_amdgcn_alignbit(v.x, left << 16, 16), __builtin_amdgcn_alignbit(v.y, v.x, 16),
                             __builtin_amdgcn_alignbit(v.z, v.y, 16), __builtin_amdgcn_alignbit(v.w, v.z, 16));
  const uint4 c = make_uint4(__builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16),
                             __builtin_amdgcn_alignbit(v.w, v.z, 16), __builtin_amdgcn_alignbit(right, v.w, 16));
  return kx == 0 ? a : (kx == 1 ? v : c);
}

// SAME: q, k and v come from one input (every transformer block; the skip connections have two): no second set of B
// operands, and the registers that frees hold a third unit per trip.
template <int WAVES, bool SAME>
__global__ __launch_bounds__(WAVES * 64) void conv_wgrad_mm_kernel(const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk,
                                                            const bf16_t* __restrict__ dv, const bf16_t* __restrict__ xq,
                                                            const bf16_t* __restrict__ xkv, float* dwq, float* dwk, float* dwv,
                                                            long long nunits, int s, float* __restrict__ part) {
  constexpr int C = 3, NW = 27;
  __shared__ float red[WAVES][2][2][256];   // [wave][q | kv][tap tile][C-layout element]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
  const int ss = s * s, upp = ss >> 5;      // 32-pixel units per patch
  constexpr bool same = SAME;
  const bool edges = s > 8;
  // this lane's taps (B operand column l15 of tap tile 0 / 1)
  int tci[2], tky[2], tkx[2]; bool tv[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = tt * 16 + l15;
    tv[tt] = t < NW;
    const int tc = tv[tt] ? t : 0;
    tci[tt] = tc / 9; tky[tt] = (tc % 9) / 3; tkx[tt] = tc % 3;
  }
  // this lane's A rows: q product row n = l15 < 3 (dq); kv product rows n < 3 (dk), 3 <= n < 6 (dv)
  const bf16_t* aq = dq + (l15 < 3 ? l15 : 0) * ss;
  const bf16_t* akv = (l15 < 3 ? dk : dv) + (l15 < 3 ? l15 : (l15 < 6 ? l15 - 3 : 0)) * ss;
  f32x4 accq[2], acckv[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) { accq[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; acckv[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  // one unit = 32 pixels of one patch; two units per trip so that twice the loads are in flight per wave (the loop is
  // bound by the latency of its dependent address -> load -> MFMA chain, not by bytes: 53 -> see DESIGN.md)
  struct UnitRegs { uint4 a_q, a_kv, bq[2], bkv[2]; unsigned lq[2], rq[2], lkv[2], rkv[2]; unsigned lm, rm[2]; };     // raw loads + their masks
  // unit -> (patch, pixel): shifts when the patch side is a power of two (8, 16, 32: every level of the presets), 32-bit
  // unsigned divisions otherwise
  const bool p2 = (s & (s - 1)) == 0;
  const int ls = 31 - __builtin_clz((unsigned)s), lupp = 2 * ls - 5;
  const unsigned nun = (unsigned)nunits;              // (launcher: nunits < 2^29)
  // Branch-free: every lane loads from a clamped, always valid address and the result is masked (AND) where the window
  // leaves the patch.  (The first version guarded each load with its own condition: ~20 exec-masked regions and 700
  // instructions per trip.)  Rows / columns of the two products that are not outputs (A rows >= 3 / 6, taps >= 27)
  // may hold anything finite: an MFMA output element depends on its own A row and B column only.
  // Edge pixels of a segment (patches wider than 8): for s = 16 / 32 the neighbouring segment of the same image row
  // is the one the adjacent lane group (lanes +-16) just loaded: two ds_bpermute instead of two 2-byte global loads.
  const bool edge_shfl = s == 16 || s == 32;
  const int lane_up = ((lane + 48) & 63) << 2, lane_dn = ((lane + 16) & 63) << 2;      // bpermute addresses: lane - 16, lane + 16
  auto load_unit = [&](unsigned uid, UnitRegs& u) {
    const bool live = uid < nun;
    const unsigned uc = live ? uid : 0u;
    unsigned patch, un;
    if (p2) { patch = uc >> lupp; un = uc & ((1u << lupp) - 1u); }
    else { patch = uc / (unsigned)upp; un = uc - patch * (unsigned)upp; }
    const int px = (int)(4 * un + lg) * 8;            // first pixel of this lane group's segment
    int y, x0;
    if (p2) { y = px >> ls; x0 = px & (s - 1); }
    else { y = (int)((unsigned)px / (unsigned)s); x0 = px - y * s; }
    const long long pbase = (long long)patch * (C * ss);
    const int seg = y * s + x0;
    u.lm = live ? 0xffffffffu : 0u;                   // units past the end contribute zero through a zero A operand
    u.a_q = *reinterpret_cast<const uint4*>(aq + pbase + seg);
    u.a_kv = *reinterpret_cast<const uint4*>(akv + pbase + seg);
    const bool has_l = x0 > 0, has_r = x0 + 8 < s;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int r = y + tky[tt] - 1;
      const unsigned rm = (r >= 0 && r < s) ? 0xffffffffu : 0u;       // zero padding above / below the patch
      const int rc = min(max(r, 0), s - 1);
      const long long off = pbase + tci[tt] * ss + rc * s + x0;
      u.rm[tt] = rm;
      u.bkv[tt] = *reinterpret_cast<const uint4*>(xkv + off);
      u.lkv[tt] = 0; u.rkv[tt] = 0; u.lq[tt] = 0; u.rq[tt] = 0;
      if (!same) u.bq[tt] = *reinterpret_cast<const uint4*>(xq + off);
      if (edges && !edge_shfl) {      // other widths: edge pixels from memory (clamped inside the row, masked at use)
        const unsigned short* ekv = reinterpret_cast<const unsigned short*>(xkv) + off;
        const unsigned short* eq = reinterpret_cast<const unsigned short*>(xq) + off;
        u.lkv[tt] = ekv[has_l ? -1 : 0]; u.rkv[tt] = ekv[has_r ? 8 : 7];
        if (!same) { u.lq[tt] = eq[has_l ? -1 : 0]; u.rq[tt] = eq[has_r ? 8 : 7]; }
        if (!has_l) { u.lkv[tt] = 0; u.lq[tt] = 0; }
        if (!has_r) { u.rkv[tt] = 0; u.rq[tt] = 0; }
      }
    }
  };
  // s = 16: a unit is two rows of two segments (lane groups 0,1 | 2,3); s = 32: one row of four
  const bool sh_l = s == 32 ? lg > 0 : (lg & 1) != 0, sh_r = s == 32 ? lg < 3 : (lg & 1) == 0;
  auto edge_l = [&](const uint4& v) { const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute(lane_up, (int)v.w) >> 16; return sh_l ? w : 0u; };
  auto edge_r = [&](const uint4& v) { const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute(lane_dn, (int)v.x) & 0xffffu; return sh_r ? w : 0u; };
  auto mma_unit = [&](const UnitRegs& u) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      // masks and shuffles here, after every load of the trip has been issued (applied at the load they made hipcc wait
      // for each load in turn)
      const unsigned rm = u.rm[tt];
      uint4 bkv = u.bkv[tt], bq = u.bq[tt];
      bkv.x &= rm; bkv.y &= rm; bkv.z &= rm; bkv.w &= rm;
      if (!same) { bq.x &= rm; bq.y &= rm; bq.z &= rm; bq.w &= rm; }
      unsigned lkv = u.lkv[tt] & rm, rkv = u.rkv[tt] & rm, lq = u.lq[tt] & rm, rq = u.rq[tt] & rm;
      if (edge_shfl) {
        lkv = edge_l(bkv); rkv = edge_r(bkv);
        if (!same) { lq = edge_l(bq); rq = edge_r(bq); }
      }
      const uint4 skv = shift_segment(bkv, lkv, rkv, tkx[tt]);
      const uint4 sq = same ? skv : shift_segment(bq, lq, rq, tkx[tt]);
      uint4 aq4 = u.a_q, akv4 = u.a_kv;
      aq4.x &= u.lm; aq4.y &= u.lm; aq4.z &= u.lm; aq4.w &= u.lm;
      akv4.x &= u.lm; akv4.y &= u.lm; akv4.z &= u.lm; akv4.w &= u.lm;
      accq[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq4), __builtin_bit_cast(bf16x8, sq), accq[tt], 0, 0, 0);
      acckv[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, akv4), __builtin_bit_cast(bf16x8, skv), acckv[tt], 0, 0, 0);
    }
  };
  const unsigned stride = gridDim.x * WAVES;
  constexpr int UPT = SAME ? 4 : 2;        // units per trip: all their loads are in flight before the first multiply
  for (unsigned uid = blockIdx.x * WAVES + wave; uid < nun; uid += UPT * stride) {
    UnitRegs u[UPT];
#pragma unroll
    for (int k = 0; k < UPT; ++k) load_unit(uid + k * stride, u[k]);         // (zero operands past the end)
#pragma unroll
    for (int k = 0; k < UPT; ++k) mma_unit(u[k]);
  }
  // C[row n = 4 lg + r][col = tap l15 (+16)]
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[wave][0][tt][(4 * lg + r) * 16 + l15] = accq[tt][r];
      red[wave][1][tt][(4 * lg + r) * 16 + l15] = acckv[tt][r];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 2 * 256; i += WAVES * 64) {
    const int which = i >> 9, tt = (i >> 8) & 1, e = i & 255;
    const int n = e >> 4, t = tt * 16 + (e & 15);
    if (t >= NW || n >= (which == 0 ? 3 : 6)) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += red[w][which][tt][e];
    if (part) { part[(long long)blockIdx.x * 1024 + i] = v; continue; }      // per-block partials, summed in block order below
    float* dst = which == 0 ? dwq + n * NW + t : (n < 3 ? dwk + n * NW + t : dwv + (n - 3) * NW + t);
    atomicAdd(dst, v);
  }
}

// deterministic tail of conv_wgrad_mm_kernel: one wave per output element i, lane l adds the partials of blocks l, l + 64, ...
// in order, then a fixed shuffle tree (a single thread walking 256 partials pays 64 dependent L2 round trips: 50 us)
__global__ __launch_bounds__(1024) void conv_wgrad_mm_reduce_kernel(const float* __restrict__ part, int nblocks, float* dwq, float* dwk,
                                                                    float* dwv) {
  constexpr int NW = 27;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int which = i >> 9, tt = (i >> 8) & 1, e = i & 255;
  const int n = e >> 4, t = tt * 16 + (e & 15);
  if (i >= 1024 || t >= NW || n >= (which == 0 ? 3 : 6)) return;          // wave-uniform
  float s[4] = {0.f, 0.f, 0.f, 0.f};      // four independent chains over loads issued together (256 blocks: one trip)
  int b = lane;
  for (; b + 3 * 64 < nblocks; b += 4 * 64) {
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = part[(long long)(b + 64 * u) * 1024 + i];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += x[u];
  }
  for (; b < nblocks; b += 64) s[0] += part[(long long)b * 1024 + i];
  float a = (s[0] + s[1]) + (s[2] + s[3]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
  if (lane == 0) {
    float* dst = which == 0 ? dwq + n * NW + t : (n < 3 ? dwk + n * NW + t : dwv + (n - 3) * NW + t);
    *dst += a;
  }
}

// VU_CONV_W=smem: the forward kernels read their weights with scalar loads inside the rolled loops (the form of rounds 1 - 3,
// kept for the A/B record: wrong under GPU sharing, see conv_fwd_kernel); default: weights staged in LDS
inline bool weights_smem() { static const bool v = [] { const char* e = getenv("VU_CONV_W"); return e && e[0] == 's'; }(); return v; }
// A/B switch for measurements: VU_CONV_SHUFFLE=0 keeps the all-loads window
inline bool shuffle_off() { static const bool v = [] { const char* e = getenv("VU_CONV_SHUFFLE"); return e && e[0] == '0'; }(); return v; }

inline int grid_for(long long items, int cap) {
  long long g = (items + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define VU_CONV_C(Cv, ...) \
  switch (Cv) { case 1: { constexpr int CC = 1; __VA_ARGS__ } break; case 2: { constexpr int CC = 2; __VA_ARGS__ } break; \
                case 3: { constexpr int CC = 3; __VA_ARGS__ } break; case 4: { constexpr int CC = 4; __VA_ARGS__ } break; \
                default: vu_set_error("conv3x3: num_channels %d not supported (1..4)", Cv); return VU_EUNSUPPORTED; }

int vu_k_conv3x3_fwd(int dtype, int out_f32, const void* in, const float* w, const float* bias, void* out,
                     long long npatch, int C, int s, hipStream_t st) {
  VU_REQUIRE(s % 4 == 0, "conv3x3: patch size must be a multiple of 4");
  VU_REQUIRE(npatch * s * s / 4 < 4294967295LL, "conv3x3: more than 2^32 pixel quads");
  const long long nq = npatch * s * s / 4;
  if (nq == 0) return VU_OK;
  const int grid = grid_for(nq, 256 * 16);
  const bool fo = out_f32 || dtype == 0;
#define VU_FWD1(WLV) \
    if (dtype == 0) hipLaunchKernelGGL((conv_fwd_kernel<float, float, CC, 1, 0, WLV>), dim3(grid), dim3(256), 0, st, (const float*)in, (const float*)in, w, w, w, bias, (float*)out, (float*)out, (float*)out, nq, s); \
    else if (fo) hipLaunchKernelGGL((conv_fwd_kernel<bf16_t, float, CC, 1, 0, WLV>), dim3(grid), dim3(256), 0, st, (const bf16_t*)in, (const bf16_t*)in, w, w, w, bias, (float*)out, (float*)out, (float*)out, nq, s); \
    else hipLaunchKernelGGL((conv_fwd_kernel<bf16_t, bf16_t, CC, 1, 0, WLV>), dim3(grid), dim3(256), 0, st, (const bf16_t*)in, (const bf16_t*)in, w, w, w, bias, (bf16_t*)out, (bf16_t*)out, (bf16_t*)out, nq, s);
  VU_CONV_C(C, if (weights_smem()) { VU_FWD1(0) } else { VU_FWD1(1) })
#undef VU_FWD1
  if (vu_prof_on()) vu_prof_note("conv_fwd_kernel<1>", 0.0, (double)nq * 4 * C * ((dtype == 0 ? 4.0 : 2.0) + (fo ? 4.0 : 2.0)));
  return vu_check_launch("vu_conv3x3_fwd");
}

int vu_k_conv3x3_qkv_fwd(int dtype, const void* xq, const void* xkv, const float* wq, const float* wk, const float* wv,
                         void* q, void* k, void* v, long long npatch, int C, int s, hipStream_t st) {
  VU_REQUIRE(s % 4 == 0, "conv3x3: patch size must be a multiple of 4");
  VU_REQUIRE(npatch * s * s / 4 < 4294967295LL, "conv3x3: more than 2^32 pixel quads");
  const long long nq = npatch * s * s / 4;
  if (nq == 0) return VU_OK;
  if (vu_conv_tz_ok(dtype, C, s, npatch)) {
    if (int e = vu_k_conv_tz_qkv_fwd(xq, xkv, wq, wk, wv, q, k, v, npatch, s, st)) return e;
    if (vu_prof_on()) vu_prof_note("conv_tz_fwd_kernel", 2.0 * 243.0 / 9.0 * C * C * nq * 4, (double)nq * 4 * C * 2.0 * (xq == xkv ? 4 : 5));
    return vu_check_launch("vu_conv3x3_qkv_fwd (Toeplitz form)");
  }
  if (vu_conv_mm_ok(dtype, C, s, 0) && npatch < 2147483647LL) {
    if (int e = vu_k_conv_mm_qkv_fwd(xq, xkv, wq, wk, wv, q, k, v, npatch, C, s, st)) return e;
    if (vu_prof_on()) vu_prof_note("conv_qkv_mm_kernel", 2.0 * 243.0 / 9.0 * C * C * nq * 4, (double)nq * 4 * C * 2.0 * (xq == xkv ? 4 : 5));
    return vu_check_launch("vu_conv3x3_qkv_fwd (MFMA form)");
  }
  const int grid = grid_for(nq, 256 * 16);
  int sh = shuffle_off() ? 0 : conv_shuffle_form(s);
  if (sh == 2) sh = 1;      // forward: VALU-bound (243 MACs per pixel), the extra DPP moves of the one-load form cost more than its loads (37.9 vs 30.9 us)
#define VU_QKV_FWD(SHV, WLV) \
    if (dtype == 0) hipLaunchKernelGGL((conv_fwd_kernel<float, float, CC, 3, SHV, WLV>), dim3(grid), dim3(256), 0, st, (const float*)xq, (const float*)xkv, wq, wk, wv, (const float*)nullptr, (float*)q, (float*)k, (float*)v, nq, s); \
    else hipLaunchKernelGGL((conv_fwd_kernel<bf16_t, bf16_t, CC, 3, SHV, WLV>), dim3(grid), dim3(256), 0, st, (const bf16_t*)xq, (const bf16_t*)xkv, wq, wk, wv, (const float*)nullptr, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, nq, s);
  VU_CONV_C(C,
    if (weights_smem()) { if (sh == 1) { VU_QKV_FWD(1, 0) } else { VU_QKV_FWD(0, 0) } }
    else { if (sh == 1) { VU_QKV_FWD(1, 1) } else { VU_QKV_FWD(0, 1) } })
#undef VU_QKV_FWD
  if (vu_prof_on()) vu_prof_note("conv_fwd_kernel<3>", 0.0, (double)nq * 4 * C * (dtype == 0 ? 4.0 : 2.0) * (xq == xkv ? 4 : 5));
  return vu_check_launch("vu_conv3x3_qkv_fwd");
}

int vu_k_conv3x3_dgrad(int dtype, int dout_f32, const void* dout, const float* w, const void* add, void* din,
                       long long npatch, int C, int s, hipStream_t st) {
  VU_REQUIRE(s % 4 == 0, "conv3x3: patch size must be a multiple of 4");
  VU_REQUIRE(npatch * s * s / 4 < 4294967295LL, "conv3x3: more than 2^32 pixel quads");
  const long long nq = npatch * s * s / 4;
  if (nq == 0) return VU_OK;
  const int grid = grid_for(nq, 256 * 16);
  VU_CONV_C(C,
    if (dtype == 0) hipLaunchKernelGGL((conv_dgrad_kernel<float, float, CC, 1>), dim3(grid), dim3(256), 0, st, (const float*)dout, (const float*)dout, (const float*)dout, w, w, w, (const float*)add, (const float*)nullptr, (float*)din, (float*)nullptr, nq, s);
    else if (dout_f32) hipLaunchKernelGGL((conv_dgrad_kernel<float, bf16_t, CC, 1>), dim3(grid), dim3(256), 0, st, (const float*)dout, (const float*)dout, (const float*)dout, w, w, w, (const bf16_t*)add, (const bf16_t*)nullptr, (bf16_t*)din, (bf16_t*)nullptr, nq, s);
    else hipLaunchKernelGGL((conv_dgrad_kernel<bf16_t, bf16_t, CC, 1>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)dout, (const bf16_t*)dout, w, w, w, (const bf16_t*)add, (const bf16_t*)nullptr, (bf16_t*)din, (bf16_t*)nullptr, nq, s);)
  if (vu_prof_on()) vu_prof_note("conv_dgrad_kernel<1>", 0.0, (double)nq * 4 * C * ((dout_f32 || dtype == 0 ? 4.0 : 2.0) + (dtype == 0 ? 4.0 : 2.0) * (add ? 2 : 1)));
  return vu_check_launch("vu_conv3x3_dgrad");
}

// dxq (and dxkv when non-null) from dq, dk, dv; add_q / add_kv are added (residual gradients).
int vu_k_conv3x3_qkv_dgrad(int dtype, const void* dq, const void* dk, const void* dv, const float* wq, const float* wk,
                           const float* wv, const void* add_q, const void* add_kv, void* dxq, void* dxkv,
                           long long npatch, int C, int s, hipStream_t st) {
  VU_REQUIRE(s % 4 == 0, "conv3x3: patch size must be a multiple of 4");
  VU_REQUIRE(npatch * s * s / 4 < 4294967295LL, "conv3x3: more than 2^32 pixel quads");
  const long long nq = npatch * s * s / 4;
  if (nq == 0) return VU_OK;
  if (vu_conv_tz_ok(dtype, C, s, npatch)) {
    if (int e = vu_k_conv_tz_qkv_dgrad(dq, dk, dv, wq, wk, wv, add_q, add_kv, dxq, dxkv, npatch, s, st)) return e;
    if (vu_prof_on()) vu_prof_note("conv_tz_dgrad_kernel", 2.0 * 243.0 / 9.0 * C * C * nq * 4, (double)nq * 4 * C * 2.0 * (dxkv ? 5 + (add_q ? 1 : 0) + (add_kv ? 1 : 0) : 4 + (add_q ? 1 : 0)));
    return vu_check_launch("vu_conv3x3_qkv_dgrad (Toeplitz form)");
  }
  if (vu_conv_mm_ok(dtype, C, s, 1) && npatch < 2147483647LL) {
    if (int e = vu_k_conv_mm_qkv_dgrad(dq, dk, dv, wq, wk, wv, add_q, add_kv, dxq, dxkv, npatch, C, s, st)) return e;
    if (vu_prof_on()) vu_prof_note("conv_qkv_dgrad_mm_kernel", 2.0 * 243.0 / 9.0 * C * C * nq * 4, (double)nq * 4 * C * 2.0 * (dxkv ? 5 + (add_q ? 1 : 0) + (add_kv ? 1 : 0) : 4 + (add_q ? 1 : 0)));
    return vu_check_launch("vu_conv3x3_qkv_dgrad (MFMA form)");
  }
  const int grid = grid_for(nq, 256 * 16);
  const int sh = shuffle_off() ? 0 : conv_shuffle_form(s);
#define VU_QKV_DGRAD(SHV) \
    if (dtype == 0) hipLaunchKernelGGL((conv_dgrad_kernel<float, float, CC, 3, SHV>), dim3(grid), dim3(256), 0, st, (const float*)dq, (const float*)dk, (const float*)dv, wq, wk, wv, (const float*)add_q, (const float*)add_kv, (float*)dxq, (float*)dxkv, nq, s); \
    else hipLaunchKernelGGL((conv_dgrad_kernel<bf16_t, bf16_t, CC, 3, SHV>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)dv, wq, wk, wv, (const bf16_t*)add_q, (const bf16_t*)add_kv, (bf16_t*)dxq, (bf16_t*)dxkv, nq, s);
  VU_CONV_C(C,
    if (sh == 2) { VU_QKV_DGRAD(2) } else if (sh == 1) { VU_QKV_DGRAD(1) } else { VU_QKV_DGRAD(0) })
#undef VU_QKV_DGRAD
  if (vu_prof_on()) vu_prof_note("conv_dgrad_kernel<3>", 0.0, (double)nq * 4 * C * (dtype == 0 ? 4.0 : 2.0) * (dxkv ? 5 + (add_q ? 1 : 0) + (add_kv ? 1 : 0) : 4 + (add_q ? 1 : 0)));
  return vu_check_launch("vu_conv3x3_qkv_dgrad");
}

// A/B switch for measurements: VU_CONV_WGRAD_VALU=1 keeps the VALU weight-gradient kernel (read once)
static bool wgrad_valu_forced() { static const bool v = getenv("VU_CONV_WGRAD_VALU") != nullptr; return v; }

static int wgrad_launch(int dtype, int dout_f32, const WgradSet& set, int nconv, long long npatch, int C, int s,
                        hipStream_t st) {
  VU_REQUIRE(s % 4 == 0, "conv3x3: patch size must be a multiple of 4");
  VU_REQUIRE(npatch * s * s / 4 < 4294967295LL, "conv3x3: more than 2^32 pixel quads");
  const long long nq = npatch * s * s / 4;
  if (nq == 0) return VU_OK;
  if (!dout_f32 && nconv == 3 && vu_conv_tz_ok(dtype, C, s, npatch) && !wgrad_valu_forced()) {      // Gram form (vu_conv_tz.hip): needs the lent slab
    void* scr = vu_deferred_take(vu_conv_tz_wgrad_scratch_floats(), st);      // the partials stay until the executor's flush: the reduce is queued
    const int defer = scr != nullptr;
    size_t scr_bytes = defer ? vu_conv_tz_wgrad_scratch_floats() * sizeof(float) : 0;
    if (!defer) vu_gemm_get_scratch(&scr, &scr_bytes);
    if (scr && scr_bytes >= vu_conv_tz_wgrad_scratch_floats() * sizeof(float)) {
      if (int e = vu_k_conv_tz_qkv_wgrad(set.dout[0], set.dout[1], set.dout[2], set.in[0], set.in[1], set.dw[0], set.dw[1], set.dw[2], (float*)scr,
                                         npatch, s, st, defer)) return e;
      if (vu_prof_on()) vu_prof_note("conv_tzw_kernel(+reduce)", 2.0 * 243.0 / 9.0 * C * C * nq * 4, (double)nq * 4 * C * (set.in[0] == set.in[1] ? 4 : 5) * 2.0);
      return vu_check_launch("vu_conv3x3_wgrad (Gram form)");
    }
  }
  if (dtype == 1 && !dout_f32 && nconv == 3 && C == 3 && s % 8 == 0 && !wgrad_valu_forced()) {
    const long long nunits = npatch * s * s / 32;
    // many waves, few blocks: every block ends in 243 float atomics on the same addresses (measured: 2048 blocks of
    // 4 waves 131 us, 512 x 4: 93 us, 256 x 4: 141 us per launch)
    constexpr int WV = 16;
    long long g = (nunits + WV - 1) / WV; if (g > 256) g = 256;
    float* part = vu_deferred_take((size_t)g * 1024, st);      // queued reduce (vu_gemm.h) when the executor has lent its arena
    const bool defer = part != nullptr;
    if (!defer) {
      void* scr = nullptr; size_t scr_bytes = 0;
      vu_gemm_get_scratch(&scr, &scr_bytes);            // lent by the model executor: deterministic sum instead of float atomics
      part = (scr && scr_bytes >= (size_t)g * 1024 * 4) ? (float*)scr : nullptr;
    }
    auto kern = set.in[0] == set.in[1] ? conv_wgrad_mm_kernel<WV, true> : conv_wgrad_mm_kernel<WV, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(WV * 64), 0, st, (const bf16_t*)set.dout[0], (const bf16_t*)set.dout[1],
                       (const bf16_t*)set.dout[2], (const bf16_t*)set.in[0], (const bf16_t*)set.in[1], set.dw[0], set.dw[1], set.dw[2], nunits, s, part);
    if (defer) {
      vu_defred d;
      memset(&d, 0, sizeof(d));
      d.kind = VU_DEFRED_WGRAD_MM; d.nblocks = (int)g; d.part = part; d.dst[0] = set.dw[0]; d.dst[1] = set.dw[1]; d.dst[2] = set.dw[2];
      vu_deferred_push(d);
    } else if (part) hipLaunchKernelGGL(conv_wgrad_mm_reduce_kernel, dim3(64), dim3(1024), 0, st, part, (int)g, set.dw[0], set.dw[1], set.dw[2]);
    if (vu_prof_on()) vu_prof_note("conv_wgrad_mm_kernel", 0.0, (double)nq * 4 * C * 5 * 2.0);
    return vu_check_launch("vu_conv3x3_wgrad");
  }
  const int gx = grid_for(nq, nconv == 1 ? 512 : 256);
  float* part = nullptr;
  bool defer = false;
  if (C == 3) {        // (the reduce kernel's 32-float rows hold 27 weights + the bias sum)
    part = vu_deferred_take((size_t)gx * nconv * C * 32, st);
    defer = part != nullptr;
    if (!defer) {
      void* scr = nullptr; size_t scr_bytes = 0;
      vu_gemm_get_scratch(&scr, &scr_bytes);            // lent by the model executor
      if (scr && scr_bytes >= (size_t)gx * nconv * C * 32 * sizeof(float)) part = (float*)scr;
    }
  }
  VU_CONV_C(C,
    if (dtype == 0) hipLaunchKernelGGL((conv_wgrad_kernel<float, float, CC>), dim3(gx, nconv * CC), dim3(256), 0, st, set, nq, s, part);
    else if (dout_f32) hipLaunchKernelGGL((conv_wgrad_kernel<float, bf16_t, CC>), dim3(gx, nconv * CC), dim3(256), 0, st, set, nq, s, part);
    else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, bf16_t, CC>), dim3(gx, nconv * CC), dim3(256), 0, st, set, nq, s, part);)
  if (defer) {
    vu_defred d;
    memset(&d, 0, sizeof(d));
    d.kind = VU_DEFRED_WGRAD3; d.nblocks = gx; d.nconv = nconv; d.part = part;
    for (int cv = 0; cv < nconv; ++cv) { d.dst[cv] = set.dw[cv]; d.dst[3 + cv] = set.dbias[cv]; }
    vu_deferred_push(d);
  } else if (part) hipLaunchKernelGGL(conv_wgrad_reduce_kernel<3>, dim3((unsigned)((nconv * 3 * 32 + 15) / 16)), dim3(1024), 0, st, set, part, gx, nconv);
  if (vu_prof_on()) vu_prof_note(nconv == 1 ? "conv_wgrad_kernel<1>" : "conv_wgrad_kernel<3>", 0.0,
                                 (double)nq * 4 * C * nconv * ((dout_f32 || dtype == 0 ? 4.0 : 2.0) + (dtype == 0 ? 4.0 : 2.0)));
  return vu_check_launch("vu_conv3x3_wgrad");
}

int vu_k_conv3x3_wgrad(int dtype, int dout_f32, const void* dout, const void* in, float* dw, float* dbias,
                       long long npatch, int C, int s, hipStream_t st) {
  WgradSet set;
  for (int i = 0; i < 3; ++i) { set.dout[i] = dout; set.in[i] = in; set.dw[i] = dw; set.dbias[i] = dbias; }
  return wgrad_launch(dtype, dout_f32, set, 1, npatch, C, s, st);
}

int vu_k_conv3x3_qkv_wgrad(int dtype, const void* dq, const void* dk, const void* dv, const void* xq, const void* xkv,
                           float* dwq, float* dwk, float* dwv, long long npatch, int C, int s, hipStream_t st) {
  WgradSet set;
  set.dout[0] = dq; set.dout[1] = dk; set.dout[2] = dv;
  set.in[0] = xq; set.in[1] = xkv; set.in[2] = xkv;
  set.dw[0] = dwq; set.dw[1] = dwk; set.dw[2] = dwv;
  set.dbias[0] = set.dbias[1] = set.dbias[2] = nullptr;
  return wgrad_launch(dtype, 0, set, 3, npatch, C, s, st);
}
