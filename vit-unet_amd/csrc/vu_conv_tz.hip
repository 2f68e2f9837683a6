// K4, Toeplitz form (round 6): the per-patch 3 x 3 q / k / v convolutions (model.py:137-139,152-154) and the sum of their data
// gradients on the matrix cores WITHOUT a tap gather.  bf16 storage, 3 channels, patch size 16 or 8.
//
// The stencil kernels of vu_conv.hip spend 243 multiply-adds per pixel on the VALU (~31 us per launch for 77 MB at 64 images:
// a third of what the same bytes cost a LayerNorm kernel); the im2col MFMA form of vu_conv_mm.hip is bound by its gather
// (eight 2-byte LDS reads per MFMA).  Here the WEIGHTS carry the stencil: along a patch row a 3-tap filter is a banded
// (Toeplitz) matrix T[xo][xi] = w[xi - xo + 1], so
//
//     out[co][y][xo] = sum_{ci, ky}  sum_xi  T_{co,ci,ky}[xo][xi] * in[ci][y + ky - 1][xi]
//
// is a plain matrix product whose B operand is the input exactly as it lies in memory: a lane's 8 k-slots are one 16-byte load
// of 8 consecutive pixels of a row, and the vertical taps are row shifts of that operand (DPP row_shr / row_shl inside a
// 16-lane row, zero shifted in = the zero halo at the patch border) or neighbouring rows of the same 16-byte loads.
//
//   s = 16: one tile = one patch.  Columns n = the 16 rows y of the patch, rows m = the 16 pixels xo of a row, K = 32 = two
//           (plane, ky) blocks of 16 input pixels.  Two 16-byte loads per lane bring the 3 planes (1.5 KB, the third twice);
//           5 k-blocks per (output plane, input tensor): (ci0 | ci1) x ky 0..2, (ci2 ky0 | ci2 ky2), (ci2 ky1 | -).
//   s = 8:  one tile = 4 patches.  Columns n = (patch, row pair), rows m = (row of the pair, xo), K = 32 = the 4 input rows
//           2j-1 .. 2j+2 of one plane x 8 pixels: the vertical taps are inside the Toeplitz block, no shifts at all;
//           3 k-blocks (one per plane) per (output plane, input tensor).
//
// A operands (the banded weight matrices, bf16 hi + lo so that the fp32 weights lose nothing that matters: the products are
// exact, the sum differs from the stencil's only in the order of the fp32 additions) are built ONCE per wave and stay in
// registers (120 / 72 VGPRs); that is why the work is split by ROLE: a wave owns one convolution (forward: q, k or v) or one
// output plane (data gradient) for its whole life.  Wave w of the launch has role w % 3 and walks tiles w / 3, + streams, ...;
// the three waves of a tile run side by side, so the input is fetched from HBM once.
//
// Weights reach the waves through LDS (vector loads, then ds_read): never through scalar loads inside a loop.
#include <stdlib.h>
#include <string.h>
#include "vu_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned tz_u4;

enum { TZ_FWD = 0, TZ_DG_SAME = 1, TZ_DG_CROSS = 2 };

struct tz_args {
  const bf16_t* in[3];    // FWD: xq, xkv, xkv;  DG: dq, dk, dv
  const float* w[3];      // wq, wk, wv  (C, C, 3, 3) fp32
  bf16_t* out[3];         // FWD: q, k, v;  DG_SAME: out[0] = dx;  DG_CROSS: out[0] = dxq, out[1] = dxkv
  const bf16_t* add[2];   // DG: residual gradients added to out[0] / out[1], or null
  int npatch, nstreams, per;      // per: tiles per stream, a multiple of the prefetch depth (4)
};

template <int CTRL, int ROWMASK>
__device__ __forceinline__ tz_u4 dpp4(const tz_u4 old, const tz_u4 v) {
  tz_u4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (unsigned)__builtin_amdgcn_update_dpp((int)old[i], (int)v[i], CTRL, ROWMASK, 0xf, true);
  return r;
}

template <int CTRL> __device__ __forceinline__ float dpp_mov_f(float v) {      // out-of-row lanes read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ f32x4 mma(const tz_u4 a, const tz_u4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const unsigned short lo = __builtin_bit_cast(unsigned short, (bf16_t)a), hi = __builtin_bit_cast(unsigned short, (bf16_t)b);
  return (unsigned)lo | ((unsigned)hi << 16);
}

// S = 16: NB = 5 k-blocks per (tensor, output); S = 8: NB = 3
template <int S, int KIND, int PF>
__global__ __launch_bounds__(256, 2) void conv_tz_kernel(const tz_args a) {
  constexpr int NB = S == 16 ? 5 : 3;
  // Roles.  FORWARD: 9 = (convolution, output channel): a wave owns ONE output plane and holds THREE bf16 terms of its weights
  // (NB operands x 3 terms = 60 / 36 registers).  hi + lo carry 16 significant bits, and q and k feed a softmax that is saturated in
  // some blocks (un-normalised skip outputs, logits ~1e4), where a 1e-5 relative change of a logit operand flips one-hot rows: with
  // two terms the teacher-forced first level-1 decoder block of Base went from 5e-3 to 9e-2 in dx; with three the products are those
  // of the fp32 weights exactly.  (Three roles of three planes each with the third term in LDS: the 15 dependent ds_read -> MFMA
  // pairs per tile cost the patch-16 forward its gain, 20 -> 28 us.)  DATA GRADIENT: 3 = the gradient's channel; two terms (a 1e-5
  // relative error of a gradient is far below its bf16 storage rounding and feeds no softmax).
  constexpr int NROLE = KIND == TZ_FWD ? 9 : 3;
  constexpr int NSL = KIND == TZ_FWD ? 1 : 3;         // slots of the operand table: the role's own plane (FWD) / the input tensors (DG)
  constexpr int NOPS = NSL * NB;
  constexpr int NT = KIND == TZ_FWD ? 1 : 3;          // input tensors a role reads
  constexpr int NO = KIND == TZ_DG_CROSS ? 2 : 1;     // accumulators (output planes) of a role
  constexpr int PE = 3 * S * S;                       // elements per patch
  __shared__ float wl[3][81];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 243; i += 256) wl[i / 81][i % 81] = a.w[i / 81][i % 81];
  __syncthreads();
  // (wave-uniform by construction; readfirstlane tells the compiler, so that pointers picked by role stay in scalar registers)
  // XCD-aware roles.  Workgroups go to the 8 XCDs round-robin and every XCD has its own L2: the NROLE roles of a tile stream must
  // sit on ONE XCD, or each of them fetches the tile from HBM again (the first mapping - consecutive waves = the roles of a stream -
  // spread them over 2 - 3 neighbouring workgroups = XCDs: rocprofv3 FETCH + WRITE 118 MB per forward launch for 77 MB, 125 for
  // the data gradient's 96).  So a whole WORKGROUP has one role, and the workgroups b, b + 8, b + 16, ... of one XCD take the roles
  // of the same four streams: role = (b / 8) % NROLE, streams 4 ((b / (8 NROLE)) 8 + b % 8) + wave.  (grid: a multiple of 8 NROLE)
  const int bid = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  const int role = (bid >> 3) % NROLE, stream = 4 * ((bid / (8 * NROLE)) * 8 + (bid & 7)) + __builtin_amdgcn_readfirstlane(wave);
  const int cv = KIND == TZ_FWD ? role / 3 : 0, cch = KIND == TZ_FWD ? role % 3 : role;      // FWD: convolution and output channel

  // ---- the role's banded weight operands, built once ----
  tz_u4 Ahi[NOPS], Alo[NOPS], A3[KIND == TZ_FWD ? NOPS : 1];
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      unsigned short hi[8], lo[8], l3[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int plane, ky, kx;
        bool ok = true;
        if constexpr (S == 16) {
          const int p = g >> 1, xi = 8 * (g & 1) + j;
          if (b < 3) { plane = p; ky = b; }
          else if (b == 3) { plane = 2; ky = p == 0 ? 0 : 2; }
          else { plane = 2; ky = 1; ok = p == 0; }
          kx = xi - l15 + 1;
        } else {
          const int r = l15 >> 3, xo = l15 & 7;
          plane = b; ky = g - r; kx = j - xo + 1;
          ok = ky >= 0 && ky <= 2;
        }
        ok = ok && kx >= 0 && kx <= 2;
        const int kyc = min(max(ky, 0), 2), kxc = min(max(kx, 0), 2);
        // FWD: role = convolution, slot = output channel, plane = input channel;  DG: slot = input tensor (= convolution), plane = its
        // channel co, role = the data-gradient channel ci, taps mirrored
        float wv;
        if constexpr (KIND == TZ_FWD) wv = wl[cv][(cch * 3 + plane) * 9 + kyc * 3 + kxc];
        else wv = wl[sl][(plane * 3 + role) * 9 + (2 - kyc) * 3 + (2 - kxc)];
        wv = ok ? wv : 0.f;
        const bf16_t h = (bf16_t)wv;
        const bf16_t l = (bf16_t)(wv - (float)h);
        const bf16_t l2 = (bf16_t)((wv - (float)h) - (float)l);
        hi[j] = __builtin_bit_cast(unsigned short, h); lo[j] = __builtin_bit_cast(unsigned short, l); l3[j] = __builtin_bit_cast(unsigned short, l2);
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        Ahi[sl * NB + b][d] = (unsigned)hi[2 * d] | ((unsigned)hi[2 * d + 1] << 16);
        Alo[sl * NB + b][d] = (unsigned)lo[2 * d] | ((unsigned)lo[2 * d + 1] << 16);
      }
      if constexpr (KIND == TZ_FWD)
        A3[sl * NB + b] = tz_u4{(unsigned)l3[0] | ((unsigned)l3[1] << 16), (unsigned)l3[2] | ((unsigned)l3[3] << 16),
                                (unsigned)l3[4] | ((unsigned)l3[5] << 16), (unsigned)l3[6] | ((unsigned)l3[7] << 16)};
    }

  const int ntiles = S == 16 ? a.npatch : (a.npatch + 3) >> 2;
  // per-lane element offsets inside a tile
  int off0, off1 = 0;
  bool rowok = true;
  int opix;                                            // output element offset of the lane's 4 pixels inside a plane of its patch
  int pin = 0;                                         // S = 8: the lane's patch inside the tile
  if constexpr (S == 16) {
    off0 = (g >> 1) * 256 + l15 * 16 + (g & 1) * 8;    // planes 0 | 1, row n = l15
    off1 = 512 + l15 * 16 + (g & 1) * 8;               // plane 2 (both halves of the wave)
    opix = l15 * 16 + 4 * g;
  } else {
    pin = l15 >> 2;
    const int jp = l15 & 3, row = 2 * jp - 1 + g;
    rowok = row >= 0 && row <= 7;
    off0 = min(max(row, 0), 7) * 8;                    // + plane * 64 (+ the lane's patch: patch_base)
    opix = (2 * jp + (g >> 1)) * 8 + 4 * (g & 1);
  }
  const bf16_t* tin[NT];
  // (selects, not a.in[role]: indexing the kernel-argument struct with a run-time index is a LOAD from the argument segment -
  // inside the tile loop it drained the whole prefetch ring with s_waitcnt vmcnt(0) once per tile)
  if constexpr (KIND == TZ_FWD) tin[0] = cv == 0 ? a.in[0] : (cv == 1 ? a.in[1] : a.in[2]);
  else { tin[0] = a.in[0]; tin[1] = a.in[1]; tin[2] = a.in[2]; }
  bf16_t* outp[NO];        // FWD: the role's tensor (channel o added below);  DG: out[o], channel `role`
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    if constexpr (KIND == TZ_FWD) outp[o] = (cv == 0 ? a.out[0] : (cv == 1 ? a.out[1] : a.out[2])) + cch * (S * S);
    else outp[o] = (o == 0 ? a.out[0] : a.out[1]) + role * (S * S);
  }
  // residual gradient of output o, channel `role`; without one the quad is still loaded (from the first input: any valid address of
  // the same extent) and masked to zero, so that the loop has no branch
  const bf16_t* addp[NO];
  unsigned admask[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const bf16_t* ad = KIND == TZ_FWD ? nullptr : (o == 0 ? a.add[0] : a.add[1]);
    addp[o] = (ad ? ad : a.in[0]) + role * (S * S);
    admask[o] = ad ? 0xffffffffu : 0u;
  }

  constexpr int NR = S == 16 ? 2 : 3;                  // raw 16-byte loads per tensor and tile
  // A tile in flight: the raw operand vectors of every tensor the role reads (+ the residual-gradient quad of the data gradient).
  // PF tiles per wave are requested ahead: the kernel is a stream of ~2 KB pieces per wave, and with one tile of prefetch and 8 waves
  // per CU only ~4 MB were in flight on the chip (2.7 TB/s at ~1.5 us of latency: the first version measured 3.0 - 3.8 TB/s).
  struct Tile { tz_u4 raw[NT][NR]; uint2 ad[NO]; };
  // NO BRANCH and NO COMPILER-TRACKED MEMORY OPERATION inside the tile loop.  hipcc's wait insertion drains the vector-memory counter
  // at every loop head (s_waitcnt vmcnt(0)), merges the counts of all paths into a join, and its scheduler gathers the refills at the
  // bottom of the body: a ring of four tiles written with plain loads ran with ONE tile in flight.  So the loads, the stores and the
  // waits of the loop are inline asm in program order (asm volatile statements keep their order), and the wait in front of a tile
  // names exactly the operations issued after that tile's loads: (PF - 1) x (loads + stores per tile) in the steady state; the
  // first PF tiles are peeled because fewer operations follow the prologue's loads.
  // Every wave runs the same `per` iterations (a multiple of PF, host-chosen so that streams x per barely exceeds the tile count);
  // an iteration past the stream's last tile REPEATS that tile (same loads, same values stored again by the same wave), and the
  // lanes of a ragged last s = 8 tile that have no patch work on the last real patch - their columns then duplicate the columns of
  // the lanes that own it, and store the same values to the same place.
  const int last = stream + ((ntiles - 1 - stream) / a.nstreams) * a.nstreams;       // (stream < ntiles: checked below)
  auto patch_base = [&](int tile) -> long long {       // element offset of the lane's patch (s = 8) / of the tile's patch (s = 16)
    if constexpr (S == 16) return (long long)tile * PE;
    else return (long long)min(tile * 4 + pin, a.npatch - 1) * PE;
  };
  constexpr int LD = NT * NR + (KIND != TZ_FWD ? NO : 0), ST = NO, OPS = LD + ST;      // vector-memory operations per tile
  auto load_tile = [&](int tile, Tile& T) {
    const long long pb = patch_base(tile);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (S == 16) {
        const bf16_t* p0 = tin[t] + pb + off0;
        const bf16_t* p1 = tin[t] + pb + off1;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(T.raw[t][0]) : "v"(p0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(T.raw[t][1]) : "v"(p1) : "memory");
      } else {
        const bf16_t* p0 = tin[t] + pb + off0;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(T.raw[t][0]) : "v"(p0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(T.raw[t][1]) : "v"(p0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(T.raw[t][2]) : "v"(p0) : "memory");
      }
    }
    if constexpr (KIND != TZ_FWD) {
#pragma unroll
      for (int o = 0; o < NO; ++o) {      // (no addend: a quad of the first input, masked to zero below)
        const bf16_t* pa = addp[o] + pb + opix;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(T.ad[o]) : "v"(pa) : "memory");
      }
    }
  };
  auto compute_tile = [&](int tile, Tile& T) {
    f32x4 acc[NO], accl[NO];            // (data gradient: the hi and lo products of an output run as two independent chains)
#pragma unroll
    for (int o = 0; o < NO; ++o) { acc[o] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[o] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 acc3 = f32x4{0.f, 0.f, 0.f, 0.f};
    const tz_u4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tz_u4 B[NB];
      if constexpr (S == 16) {
        B[0] = dpp4<0x111, 0xf>(z4, T.raw[t][0]);                                   // row_shr:1: lane y takes row y - 1 (ky = 0)
        B[1] = T.raw[t][0];
        B[2] = dpp4<0x101, 0xf>(z4, T.raw[t][0]);                                   // row_shl:1: row y + 1 (ky = 2)
        B[3] = dpp4<0x101, 0xc>(dpp4<0x111, 0x3>(z4, T.raw[t][1]), T.raw[t][1]);    // k-slots 0..15: plane 2 ky 0; 16..31: plane 2 ky 2
        B[4] = T.raw[t][1];
      } else {
        const unsigned m = rowok ? 0xffffffffu : 0u;
#pragma unroll
        for (int c = 0; c < 3; ++c) B[c] = tz_u4{T.raw[t][c][0] & m, T.raw[t][c][1] & m, T.raw[t][c][2] & m, T.raw[t][c][3] & m};
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if constexpr (KIND == TZ_FWD) {      // three independent chains (hi, lo, third term), added at the end
          acc[0] = mma(Ahi[b], B[b], acc[0]);
          accl[0] = mma(Alo[b], B[b], accl[0]);
          acc3 = mma(A3[b], B[b], acc3);
        } else {
          const int o = (KIND == TZ_DG_CROSS && t > 0) ? 1 : 0;
          acc[o] = mma(Ahi[t * NB + b], B[b], acc[o]);
          if constexpr (KIND == TZ_DG_CROSS && S == 16) acc[o] = mma(Alo[t * NB + b], B[b], acc[o]);      // (one chain: this instantiation sits at 256 registers)
          else accl[o] = mma(Alo[t * NB + b], B[b], accl[o]);
        }
      }
    }
    // ---- epilogue: (+ residual gradient), round once, 8-byte stores ----
    const long long ob = patch_base(tile) + opix;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      f32x4 v = acc[o];
      if constexpr (KIND == TZ_FWD) v += accl[o] + acc3;
      if constexpr (KIND != TZ_FWD) {
        v += accl[o];
        const uint2 r = make_uint2(T.ad[o].x & admask[o], T.ad[o].y & admask[o]);
        v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
        v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
      }
      const uint2 pk = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
      bf16_t* dst = outp[o] + ob;
      asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(pk) : "memory");
    }
  };
  // the tile's registers are outputs of asm loads that may still be in flight: nothing may read them above this point, so they pass
  // through the wait as read-write operands
  auto tie = [&](Tile& T) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < NR; ++c) asm volatile("" : "+v"(T.raw[t][c]));
    if constexpr (KIND != TZ_FWD) {
#pragma unroll
      for (int o = 0; o < NO; ++o) asm volatile("" : "+v"(T.ad[o]));
    }
  };

  if (stream >= ntiles) return;          // (wave-uniform; only launches smaller than their stream count)
  Tile ring[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) load_tile(min(stream + u * a.nstreams, last), ring[u]);
#define TZ_STEP(U, NWAIT, K)                                                                      \
  {                                                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NWAIT) : "memory");                               \
    tie(ring[U]);                                                                                 \
    compute_tile(min(stream + ((K) + (U)) * a.nstreams, last), ring[U]);                          \
    load_tile(min(stream + ((K) + (U) + PF) * a.nstreams, last), ring[U]);                        \
  }
  // peeled first round: after the loads of ring[i] came (PF - 1 - i) tiles of prologue loads and i whole iterations
  TZ_STEP(0, (PF - 1) * LD, 0)
  TZ_STEP(1, (PF - 2) * LD + OPS, 0)
  if constexpr (PF > 2) {
    TZ_STEP(2, (PF - 3) * LD + 2 * OPS, 0)
    TZ_STEP(3, 3 * OPS, 0)
  }
  for (int k = PF; k < a.per; k += PF) {
    TZ_STEP(0, (PF - 1) * OPS, k)
    TZ_STEP(1, (PF - 1) * OPS, k)
    if constexpr (PF > 2) {
      TZ_STEP(2, (PF - 1) * OPS, k)
      TZ_STEP(3, (PF - 1) * OPS, k)
    }
  }
#undef TZ_STEP
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");      // (the last refills land in registers nobody reads: let them, before the wave ends)
}

// ---- weight gradients: a Gram product over the pixels of a row -------------------------------------------------------------------
// dW[co][ci][ky][kx] = sum_{patch, y, x} dOut[co][y][x] * in[ci][y + ky - 1][x + kx - 1]  (model.py:137-139 backward).  With the PIXELS of
// a row as the contraction index both operands are again the tensors as they lie in memory (a lane's 8 k-slots = one 16-byte load):
//
//     G_{co,ci,kx}[y][y'] = sum_{patch, x} dOut[co][y][x] * in[ci][y'][x + kx - 1]          (16 x 16, accumulated over ALL tiles)
//     dW[co][ci][ky][kx]  = sum_y G_{co,ci,kx}[y][y + ky - 1]                                 (three diagonals, once, at the end)
//
// s = 16: K = 32 = 2 patches x 16 pixels (a tile = 2 patches); s = 8: K = 32 = 4 patches x 8 pixels and the 16 rows / columns are
// (patch group h, y): a tile = 8 patches, only the h = h' blocks of G are read.  The horizontal tap is a 16-bit funnel shift of the
// input operand (s = 16: the pixel that crosses the 8-pixel chunk comes from the neighbouring 16-lane row by ds_bpermute); the vertical
// tap costs nothing until the final diagonal sums.  A wave owns one convolution (role = q, k or v: 3 dOut planes x 3 input planes x
// 3 kx = 27 accumulators = 108 registers) for its whole life and reads 6 KB per tile: every byte of dq, dk, dv and x is fetched from
// HBM once (the im2col form of vu_conv.hip issued 4 KB of loads per 32 pixels: every tap its own copy of the rows).
// Per-workgroup partial sums go to a slab lent by the caller and are added in block order by conv_tzw_reduce_kernel: deterministic.
struct tzw_args {
  const bf16_t* dout[3];   // dq, dk, dv
  const bf16_t* x[3];      // xq, xkv, xkv
  float* part;             // [gridDim.x][96] partial sums (81 used)
  int npatch, nstreams, per;
};

template <int S>
__global__ __launch_bounds__(256, 2) void conv_tzw_kernel(const tzw_args a) {
  constexpr int PE = 3 * S * S, TP = S == 16 ? 2 : 8, PF = 2, LD = 6;
  __shared__ float red[81][17];          // [weight][wave x 16-lane row] (+1: the 81 final readers walk different rows)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  // (XCD-aware like conv_tz_kernel: the three roles of a stream in workgroups b, b + 8, b + 16 of one XCD - they share x in its L2)
  const int role = (blockIdx.x >> 3) % 3, stream = 4 * ((blockIdx.x / 24) * 8 + (blockIdx.x & 7)) + wave;          // (wave-uniform)
  const bf16_t* dp = role == 0 ? a.dout[0] : (role == 1 ? a.dout[1] : a.dout[2]);
  const bf16_t* xp = role == 0 ? a.x[0] : (role == 1 ? a.x[1] : a.x[2]);
  const int ntiles = (a.npatch + TP - 1) / TP;
  int pin, off;
  if constexpr (S == 16) { pin = g >> 1; off = l15 * 16 + (g & 1) * 8; }
  else { pin = 4 * (l15 >> 3) + g; off = (l15 & 7) * 8; }
  struct Tile { tz_u4 A[3], X[3]; };
  auto load_tile = [&](int tile, Tile& T) {
    const long long pb = (long long)min(tile * TP + pin, a.npatch - 1) * PE + off;
    const bf16_t* pa = dp + pb;
    const bf16_t* px = xp + pb;
    if constexpr (S == 16) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(T.A[0]) : "v"(pa) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(T.A[1]) : "v"(pa) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(T.A[2]) : "v"(pa) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(T.X[0]) : "v"(px) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(T.X[1]) : "v"(px) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(T.X[2]) : "v"(px) : "memory");
    } else {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(T.A[0]) : "v"(pa) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(T.A[1]) : "v"(pa) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(T.A[2]) : "v"(pa) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(T.X[0]) : "v"(px) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(T.X[1]) : "v"(px) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(T.X[2]) : "v"(px) : "memory");
    }
  };
  f32x4 acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane_up = ((lane + 48) & 63) << 2, lane_dn = ((lane + 16) & 63) << 2;      // bpermute byte addresses of lanes - 16 / + 16
  const bool c1 = (g & 1) != 0;
  auto compute_tile = [&](int tile, const Tile& T) {
    // a patch past the end (ragged last tile) contributes nothing: its dOut operand is zeroed (its loads were clamped to a real patch)
    const unsigned am = tile * TP + pin < a.npatch ? 0xffffffffu : 0u;
    tz_u4 B[3][3];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const tz_u4 v = T.X[ci];
      unsigned l = 0u, r = 0u;
      if constexpr (S == 16) {
        const unsigned lw = (unsigned)__builtin_amdgcn_ds_bpermute(lane_up, (int)v[3]) >> 16;      // pixel 7 of the row, held by the chunk-0 lane
        const unsigned rw = (unsigned)__builtin_amdgcn_ds_bpermute(lane_dn, (int)v[0]) & 0xffffu;  // pixel 8, held by the chunk-1 lane
        l = c1 ? lw : 0u; r = c1 ? 0u : rw;
      }
      B[ci][0] = tz_u4{__builtin_amdgcn_alignbit(v[0], l << 16, 16), __builtin_amdgcn_alignbit(v[1], v[0], 16),
                       __builtin_amdgcn_alignbit(v[2], v[1], 16), __builtin_amdgcn_alignbit(v[3], v[2], 16)};      // slot j = pixel j - 1
      B[ci][1] = v;
      B[ci][2] = tz_u4{__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                       __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(r, v[3], 16)};            // slot j = pixel j + 1
    }
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const tz_u4 A = tz_u4{T.A[co][0] & am, T.A[co][1] & am, T.A[co][2] & am, T.A[co][3] & am};
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[(co * 3 + ci) * 3 + kx] = mma(A, B[ci][kx], acc[(co * 3 + ci) * 3 + kx]);
    }
  };
  auto tie = [&](Tile& T) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { asm volatile("" : "+v"(T.A[c])); asm volatile("" : "+v"(T.X[c])); }
  };
  if (stream < ntiles) {
    const int last = stream + ((ntiles - 1 - stream) / a.nstreams) * a.nstreams;
    const int cnt = (ntiles - 1 - stream) / a.nstreams + 1;                       // this stream's tiles; iterations past them add nothing
    Tile ring[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) load_tile(min(stream + u * a.nstreams, last), ring[u]);
    // (same discipline as conv_tz_kernel: asm loads and waits in program order, no branch in the loop; a repeated last tile is
    // switched off through the dOut mask: tile index past the end -> `am` = 0 via an out-of-range patch number)
#define TZW_STEP(U, NWAIT, K)                                                                        \
    {                                                                                                \
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NWAIT) : "memory");                                \
      tie(ring[U]);                                                                                  \
      const int kk_ = (K) + (U);                                                                     \
      compute_tile(kk_ < cnt ? stream + kk_ * a.nstreams : 0x3fffffff / TP, ring[U]);                \
      load_tile(min(stream + (kk_ + PF) * a.nstreams, last), ring[U]);                               \
    }
    TZW_STEP(0, (PF - 1) * LD, 0)
    TZW_STEP(1, (PF - 1) * LD, 0)
    for (int k = PF; k < a.per; k += PF) {
      TZW_STEP(0, (PF - 1) * LD, k)
      TZW_STEP(1, (PF - 1) * LD, k)
    }
#undef TZW_STEP
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
  }
  // ---- the three diagonals of every G: acc[r] of lane (l15, g) is G[y = 4 g + r][y' = l15] ----
  // Per lane the 81 diagonal contributions are plain selects; a 16-lane row sums them with four DPP adds (VALU only), the row totals
  // of the 4 x 4 (wave, row) pairs go through LDS and 81 threads add them in a fixed order.  (A 64-lane shuffle tree per value - 486
  // ds_bpermute per wave through the CU's one LDS crossbar - cost this kernel 13 us; lane slots added wave after wave 6 us.)
#pragma unroll
  for (int i = 0; i < 27; ++i)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * g + r;
        bool on;
        if constexpr (S == 16) on = l15 == m + ky - 1;
        else on = (l15 >> 3) == (m >> 3) && (l15 & 7) == (m & 7) + ky - 1;
        v += on ? acc[i][r] : 0.f;
      }
      v += dpp_mov_f<0x111>(v); v += dpp_mov_f<0x112>(v); v += dpp_mov_f<0x114>(v); v += dpp_mov_f<0x118>(v);      // row_shr:1,2,4,8: lane 15 = row total
      if (l15 == 15) red[(i / 3) * 9 + ky * 3 + (i % 3)][wave * 4 + g] = v;          // (co, ci) x 9 + ky x 3 + kx
    }
  __syncthreads();
  if (threadIdx.x < 81) {
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 16; ++l) s4[l & 3] += red[threadIdx.x][l];
    a.part[(long long)blockIdx.x * 96 + threadIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
}

// one wave per weight: the partial sums of the role's workgroups in a fixed order, then a fixed shuffle tree
__global__ __launch_bounds__(1024) void conv_tzw_reduce_kernel(const float* __restrict__ part, int nblocks, float* dwq, float* dwk, float* dwv) {
  const int o = blockIdx.x * 16 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= 243) return;
  const int role = o / 81, e = o % 81;
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  // the role's workgroups: b with (b / 8) % 3 == role, i.e. b = 24 i + 8 role + x (x = 0..7); lane l takes the n-th of them, n = l, l + 64, ...
  const int nrole = nblocks / 3;
  int u = 0;
  for (int n = lane; n < nrole; n += 64, u = (u + 1) & 3) {
    const int b = 24 * (n >> 3) + 8 * role + (n & 7);
    s4[u] += part[(long long)b * 96 + e];
  }
  float v = (s4[0] + s4[1]) + (s4[2] + s4[3]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  if (lane == 0) { float* dst = role == 0 ? dwq : (role == 1 ? dwk : dwv); dst[e] += v; }
}

inline int tz_mode() {      // VU_CONV_TZ=0: the stencil kernels of vu_conv.hip (A/B record); default on
  static const int v = [] { const char* e = getenv("VU_CONV_TZ"); return e ? (e[0] == '0' ? 0 : 1) : 1; }();
  return v;
}

inline int tz_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
template <int S, int KIND>
int tz_launch(tz_args& a, hipStream_t st) {
  const int ntiles = S == 16 ? a.npatch : (a.npatch + 3) / 4;
  // 2 workgroups of 4 waves per CU; streams (tile walkers) in multiples of 4 so that 3 roles x streams fills whole workgroups
  // `per` tiles per stream (a multiple of 4 = the deepest prefetch ring), then as few streams as cover the tiles: 12544 tiles ->
  // per 20, 628 streams (16 tile slots repeat a tile)
  // measured (tools/conv_bench.py, 64 and 16 images): two or four tiles of prefetch are equal; 680 streams (two workgroups per CU) for
  // large launches, 340 (one per CU: half the per-wave set-up) below ~8 tiles per stream; 1020 streams are slower everywhere
  static const int pf_env = tz_env("VU_TZ_PF", 0), ns_env = tz_env("VU_TZ_NS", 0);
  const int pf = pf_env ? pf_env : 2;
  // (forward: 9 roles per stream - a third of the streams fill the same wave slots)
  constexpr int NROLE = KIND == TZ_FWD ? 9 : 3;
  const int nsc = ns_env ? ns_env : (KIND == TZ_FWD ? 226 : (ntiles >= 5440 ? 680 : 340));      // (forward, 9 roles: 226 streams fill the wave slots at every size measured)
  int per = ((ntiles + nsc - 1) / nsc + pf - 1) / pf * pf;
  int ns = ((ntiles + per - 1) / per + 31) / 32 * 32;          // streams in groups of 32 = 8 XCDs x the 4 waves of a workgroup
  a.nstreams = ns; a.per = per;
  const unsigned grid = (unsigned)(ns / 4 * NROLE);
  if (pf == 4) {
    if constexpr (KIND == TZ_FWD) hipLaunchKernelGGL((conv_tz_kernel<S, KIND, 4>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_tz_kernel<S, KIND, 2>), dim3(grid), dim3(256), 0, st, a);
  } else hipLaunchKernelGGL((conv_tz_kernel<S, KIND, 2>), dim3(grid), dim3(256), 0, st, a);
  return VU_OK;
}

}  // namespace

bool vu_conv_tz_ok(int dtype, int C, int s, long long npatch) {
  return tz_mode() == 1 && dtype == 1 && C == 3 && (s == 16 || s == 8) && npatch > 0 && npatch < (1LL << 24);
}

int vu_k_conv_tz_qkv_fwd(const void* xq, const void* xkv, const float* wq, const float* wk, const float* wv, void* q, void* k, void* v,
                         long long npatch, int s, hipStream_t st) {
  tz_args a;
  a.in[0] = (const bf16_t*)xq; a.in[1] = a.in[2] = (const bf16_t*)xkv;
  a.w[0] = wq; a.w[1] = wk; a.w[2] = wv;
  a.out[0] = (bf16_t*)q; a.out[1] = (bf16_t*)k; a.out[2] = (bf16_t*)v;
  a.add[0] = a.add[1] = nullptr;
  a.npatch = (int)npatch;
  return s == 16 ? tz_launch<16, TZ_FWD>(a, st) : tz_launch<8, TZ_FWD>(a, st);
}

int vu_k_conv_tz_qkv_dgrad(const void* dq, const void* dk, const void* dv, const float* wq, const float* wk, const float* wv,
                           const void* add_q, const void* add_kv, void* dxq, void* dxkv, long long npatch, int s, hipStream_t st) {
  tz_args a;
  a.in[0] = (const bf16_t*)dq; a.in[1] = (const bf16_t*)dk; a.in[2] = (const bf16_t*)dv;
  a.w[0] = wq; a.w[1] = wk; a.w[2] = wv;
  a.out[0] = (bf16_t*)dxq; a.out[1] = (bf16_t*)dxkv; a.out[2] = nullptr;
  a.add[0] = (const bf16_t*)add_q; a.add[1] = (const bf16_t*)add_kv;
  a.npatch = (int)npatch;
  if (dxkv) return s == 16 ? tz_launch<16, TZ_DG_CROSS>(a, st) : tz_launch<8, TZ_DG_CROSS>(a, st);
  return s == 16 ? tz_launch<16, TZ_DG_SAME>(a, st) : tz_launch<8, TZ_DG_SAME>(a, st);
}

size_t vu_conv_tz_wgrad_scratch_floats() { return (size_t)510 * 96; }
int vu_k_conv_tz_qkv_wgrad(const void* dq, const void* dk, const void* dv, const void* xq, const void* xkv, float* dwq, float* dwk, float* dwv,
                           float* part, long long npatch, int s, hipStream_t st, int defer) {
  tzw_args a;
  a.dout[0] = (const bf16_t*)dq; a.dout[1] = (const bf16_t*)dk; a.dout[2] = (const bf16_t*)dv;
  a.x[0] = (const bf16_t*)xq; a.x[1] = a.x[2] = (const bf16_t*)xkv;
  a.part = part; a.npatch = (int)npatch;
  const int tp = s == 16 ? 2 : 8, ntiles = (int)((npatch + tp - 1) / tp);
  // streams PER ROLE (every role walks all tiles): 170 workgroups x 4 waves for large launches, 85 x 4 below ~8 tiles per stream
  const int wgs = ntiles >= 5440 ? 168 : (ntiles >= 340 ? 88 : ((ntiles + 3) / 4 + 7) / 8 * 8);      // (multiples of 8: one per XCD)
  a.nstreams = wgs * 4;
  a.per = ((ntiles + a.nstreams - 1) / a.nstreams + 1) / 2 * 2;
  if (s == 16) hipLaunchKernelGGL(conv_tzw_kernel<16>, dim3((unsigned)(3 * wgs)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv_tzw_kernel<8>, dim3((unsigned)(3 * wgs)), dim3(256), 0, st, a);
  if (defer) {
    vu_defred d;
    memset(&d, 0, sizeof(d));
    d.kind = VU_DEFRED_TZW; d.nblocks = 3 * wgs; d.part = part; d.dst[0] = dwq; d.dst[1] = dwk; d.dst[2] = dwv;
    vu_deferred_push(d);
  } else
  hipLaunchKernelGGL(conv_tzw_reduce_kernel, dim3(16), dim3(1024), 0, st, (const float*)part, 3 * wgs, dwq, dwk, dwv);
  return VU_OK;
}
