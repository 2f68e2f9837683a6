// The plain big bf16 products of levels 0 / 1 - the 3072 x 3072 and 768 x 768 Linear layers (model.py:102-108,147-148,162-163):
// forward y = x W^T + b (+ projection dropout + block residual), data gradient dx = dy W, weight gradient dW += dy^T x - on a
// kernel built for ONE workgroup per CU (round 4; rounds 1 - 3 ran them on vu_gemm.h's 128 x 128 register-staged tile at
// 0.45 PFLOP/s, round 3 handed them to hipBLASLt).
//
//   * 512 threads = 8 waves as 2 (M) x 4 (N); wave tile (TM x 16) x (TN x 16) of v_mfma_f32_16x16x32_bf16 tiles.  The shapes of
//     this model decide the block tile: 3136 = 14 x 224 token rows (64 images x 49) and 12544 = 56 x 224, 3072 = 16 x 192 and
//     768 = 4 x 192 columns, so 224 x 192 (TM = 7, TN = 3) covers every level-0 / level-1 product at 64 images with 224 exact
//     tiles: one round on 256 CUs at 87.5 % occupancy (256 x 256 tiles would leave 156 / 147 tiles, 61 % / 57 %).
//   * K-steps of 64 through a 3-stage LDS ring (52 KB per stage) filled by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, no LDS store instructions); ONE s_barrier per k-step: before it every wave waits (counted vmcnt) for its
//     own pieces of stage t, behind it stage t is complete and stage t - 1 is free, so the loads of step t + 2 are issued
//     there and stay in flight under the matrix instructions of steps t and t + 1.  The DMA instructions are inline asm so
//     that hipcc does not drain them (it waits vmcnt(0) before any ds_read while a builtin LDS-DMA is outstanding).
//   * An LDS-DMA writes 1 KiB contiguously (wave base + lane x 16), so the images are linear and the bank-conflict swizzles sit
//     on the SOURCE address: k-contiguous operands ([row][64] images, 128-byte rows, ds_read_b128 fragments) XOR the 16-byte
//     chunk with row & 7; row-contiguous operands ([64][rows] images, ds_read_b64_tr_b16 fragments) XOR it with a function of k
//     that spreads the eight k-rows a 32-lane half reads over all 64 banks (three pitch classes, below).
//   * Epilogue through LDS (fp32, one half of the tile rows per pass): bias, dropout (same mask as vu_gemm's epilogue:
//     element index m N + n through vu_keep), residual addend, all as 16-byte accesses of whole 384-byte row segments; the
//     fp32-accumulating form (weight gradients) adds into C the same way.
//   * Tiles are dealt to the XCDs in rectangles (blocks id and id + 8 share an XCD's L2): 7 x 4 tiles per XCD at 14 x 16.
// Operand forms (original orientation, C = A B): A (M, K) k-contiguous or m-contiguous, B (K, N) k-contiguous (a Linear
// weight (N, K)) or n-contiguous.  K % 64 == 0; M, N, leading dimensions % 8 == 0; 16-byte aligned bases.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "vu_gemm.h"
#include "vu_kernels.h"

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int BK = 64;

// one LDS-DMA piece: 64 lanes x 16 bytes from sbase + voff[lane] to LDS bytes [lds_dst, lds_dst + 1024) in lane order.
// M0 carries the LDS address and is compiler-reserved: saved and restored inside the statement.  Not counted by hipcc:
// every wait for these loads is an explicit s_waitcnt vmcnt below.
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_dst) : "memory");
}
// the same for 2 - 4 pieces whose LDS destinations are STRIDE bytes apart (pieces wave, wave + NW, ... of a stage): one save / restore
// of M0, one s_add per further piece
template <int STRIDE> __device__ __forceinline__ void dma16x2(const void* b0, unsigned v0, const void* b1, unsigned v1, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
               "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(b0), "v"(v0), "s"(b1), "v"(v1), "s"(lds_dst), "n"(STRIDE) : "memory", "scc");
}
template <int STRIDE> __device__ __forceinline__ void dma16x3(const void* b0, unsigned v0, const void* b1, unsigned v1, const void* b2, unsigned v2, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
               "s_add_u32 m0, m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %3\n\t"
               "s_add_u32 m0, m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %5\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(b0), "v"(v0), "s"(b1), "v"(v1), "s"(b2), "v"(v2), "s"(lds_dst), "n"(STRIDE) : "memory", "scc");
}
template <int STRIDE> __device__ __forceinline__ void dma16x4(const void* b0, unsigned v0, const void* b1, unsigned v1, const void* b2, unsigned v2, const void* b3,
                                        unsigned v3, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
               "s_add_u32 m0, m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %3\n\t"
               "s_add_u32 m0, m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %5\n\t"
               "s_add_u32 m0, m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %7\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(b0), "v"(v0), "s"(b1), "v"(v1), "s"(b2), "v"(v2), "s"(b3), "v"(v3), "s"(lds_dst), "n"(STRIDE) : "memory", "scc");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// chunk XOR of a row-contiguous ([k][BX]) image: the transposing read of one 32-lane half takes 8 bytes x 4 lanes from each
// of the k-rows 8 g + q and 8 (g + 1) + q (q = 0..3): eight 32-byte spans that must fall on eight different 8-bank slots.
//   pitch % 256 == 0   (BX % 128 == 0): every row starts on the same banks     -> 3 bits: (k & 3) | ((k >> 3) & 1) << 2
//   pitch % 256 == 128 (BX % 64 == 0):  rows alternate between two half-rows    -> 2 bits: ((k >> 1) & 1) | ((k >> 3) & 1) << 1
//   pitch % 128 == 64  (BX % 32 == 0):  rows q = 0..3 already take four slots   -> 1 bit:  (k >> 3) & 1
// in units of 32 bytes (two chunks), so the two chunks of a span stay together.
template <int BX> __device__ __forceinline__ int tswz(int k) {
  if constexpr (BX % 128 == 0) return ((k & 3) | (((k >> 3) & 1) << 2)) << 1;
  else if constexpr (BX % 64 == 0) return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1;
  else return ((k >> 3) & 1) << 1;
}

struct bg_args {
  const bf16_t* A; const bf16_t* B; void* C;
  int M, N, K;
  long long lda, ldb, ldc;          // leading dimensions of the stored matrices (elements)
  const float* bias;                // [N] or null
  const bf16_t* addend;             // C layout or null
  int dropout; vu_rng rng;
  int accumulate;                   // fp32 C: C += product
  int tiles_m, tiles_n, gm, gn;     // tile grid and its split into XCD rectangles (gm * gn == 8, or 0: identity order)
};

// TA: A stored (K, M) (m-contiguous); TB: B stored (K, N) (n-contiguous); otherwise k-contiguous (M, K) / (N, K).
// WM: wave rows (2: 512 threads, one workgroup per CU; 1: 256 threads, two per CU when NST = 2); NST: LDS ring stages
// WN: wave columns (4: eight waves of 112 x 48 or four of 112 x 48 / 112 x 32; 2: FOUR waves of 112 x 96, one per SIMD - half the
// waves re-reading each fragment, 35 % fewer LDS bytes per product, accumulators in the AGPR half of a 512-register budget)
template <bool TA, bool TB, bool CF, int TM, int TN, int WM, int NST, int WN = 4>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && TM * TN > 24) ? 1 : 2) void bgemm_kernel(const bg_args g) {
  constexpr int NW = WM * WN, BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int NPA = BM / 8, NPB = BN / 8, NP = NPA + NPB;               // 1-KiB pieces per stage
  constexpr int STAGE = NP * 1024;
  constexpr int NJ = (NP + NW - 1) / NW;                                   // pieces per wave (the last one only for waves < NP % NW)
  static_assert(NST * STAGE <= 160 * 1024, "LDS ring too large");
  static_assert((TA ? BM % 32 == 0 : BM % 8 == 0) && (TB ? BN % 32 == 0 : BN % 8 == 0), "tile extents: multiples of 32 (row-contiguous operand) / 8");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lbase = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;          // (WM = 1: wm = 0)
  // ---- tile of this workgroup (XCD rectangles) -------------------------------------------------------------------
  int tile_m, tile_n;
  {
    const int bid = blockIdx.x;
    if (g.gm) {
      const int x = bid & 7, idx = bid >> 3, rm = x / g.gn, rn = x % g.gn, pm = g.tiles_m / g.gm, pn = g.tiles_n / g.gn;
      tile_m = rm * pm + idx / pn; tile_n = rn * pn + idx % pn;
    } else { tile_m = bid / g.tiles_n; tile_n = bid % g.tiles_n; }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // ---- per-piece source offsets (bytes from the operand base of the k-step) -------------------------------------------
  // kvalid: k-extent of the step (64, or K % 64 in the last step of a ragged K): k-rows / k-chunks beyond it are fetched from the
  // last valid ones (in-bounds, finite) and multiplied by zeroed A fragments in the tile loop
  constexpr int NJF = NP / NW, NREM = NP % NW;     // every wave issues NJF pieces per stage, waves < NREM one more
  const bool has_last = NREM != 0 && wave < NREM;   // (wave-uniform)
  bool isA[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) isA[j] = wave + NW * j < NPA;
  auto piece_off = [&](int j, int kvalid) __attribute__((always_inline)) -> unsigned {
    const int p = wave + NW * j;
    const int pp = p < NP ? p : NP - 1;
    const bool a = pp < NPA;
    const int q = a ? pp : pp - NPA;                 // piece within its operand's image
    const int o = q * 1024 + lane * 16;              // byte offset in the (linear) image
    const int X = a ? g.M : g.N, x0 = a ? m0 : n0;
    long long e;
    if (a ? !TA : !TB) {                             // [row][64]: 8 rows of 128 bytes per piece
      const int r = o >> 7, pc = (o >> 4) & 7, lc = pc ^ (r & 7);
      int row = x0 + r; row = row < X ? row : X - 1;
      int kk = lc * 8; kk = kk < kvalid ? kk : kvalid - 8;
      e = (long long)row * (a ? g.lda : g.ldb) + kk;
    } else {                                         // [64][BX]
      const int BX = a ? BM : BN, pitch = BX * 2;
      int k = o / pitch;
      const int pc = (o - k * pitch) >> 4;
      const int lc = pc ^ (a ? tswz<BM>(k) : tswz<BN>(k));
      int col = x0 + lc * 8; col = col + 8 <= X ? col : X - 8;
      k = k < kvalid ? k : kvalid - 1;
      e = (long long)k * (a ? g.lda : g.ldb) + col;
    }
    return (unsigned)(e * 2);
  };
  unsigned voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) voff[j] = piece_off(j, BK);
  const int nk = (g.K + BK - 1) / BK, ktail = g.K - (nk - 1) * BK;          // ktail == 64: no ragged step
  const long long stepA = TA ? (long long)BK * g.lda * 2 : (long long)BK * 2;      // bytes per k-step
  const long long stepB = TB ? (long long)BK * g.ldb * 2 : (long long)BK * 2;
  const char* baseA = reinterpret_cast<const char*>(g.A);
  const char* baseB = reinterpret_cast<const char*>(g.B);
  // piece j of k-step `step` (j = NJ - 1: only the waves that have one)
  auto issue_piece = [&](int step, int j) __attribute__((always_inline)) {
    if (j == NJ - 1 && NREM != 0 && !has_last) return;
    const unsigned st = lbase + (unsigned)(step % NST) * STAGE + (unsigned)(wave + NW * j) * 1024;
    const char* sb = (isA[j] ? baseA + step * stepA : baseB + step * stepB);
    if (step == nk - 1 && ktail != BK) dma16(sb, piece_off(j, ktail), st);           // (workgroup-uniform)
    else dma16(sb, voff[j], st);
  };
  // all pieces of a k-step: batches of up to four pieces per asm statement
  auto issue = [&](int step) __attribute__((always_inline)) {
    if (step == nk - 1 && ktail != BK) {             // (workgroup-uniform; ragged last step: clamped offsets, piece by piece)
#pragma unroll
      for (int j = 0; j < NJ; ++j) issue_piece(step, j);
      return;
    }
    const char* sa = baseA + step * stepA;
    const char* sb = baseB + step * stepB;
    const unsigned st = lbase + (unsigned)(step % NST) * STAGE + (unsigned)wave * 1024;
    auto bs = [&](int j) __attribute__((always_inline)) -> const char* { return isA[j] ? sa : sb; };
    constexpr int NFULL = NREM ? NJ - 1 : NJ;          // pieces every wave has
    constexpr int SB = NW * 1024;                       // LDS bytes between a wave's consecutive pieces
    int j = 0;
#pragma unroll
    for (; j + 4 <= NFULL; j += 4) dma16x4<SB>(bs(j), voff[j], bs(j + 1), voff[j + 1], bs(j + 2), voff[j + 2], bs(j + 3), voff[j + 3], st + j * SB);
    if constexpr (NFULL % 4 == 3) dma16x3<SB>(bs(NFULL - 3), voff[NFULL - 3], bs(NFULL - 2), voff[NFULL - 2], bs(NFULL - 1), voff[NFULL - 1], st + (NFULL - 3) * SB);
    if constexpr (NFULL % 4 == 2) dma16x2<SB>(bs(NFULL - 2), voff[NFULL - 2], bs(NFULL - 1), voff[NFULL - 1], st + (NFULL - 2) * SB);
    if constexpr (NFULL % 4 == 1) dma16(bs(NFULL - 1), voff[NFULL - 1], st + (NFULL - 1) * SB);
    if (NREM != 0 && has_last) dma16(bs(NJ - 1), voff[NJ - 1], st + (NJ - 1) * SB);
  };
  // wait until at most `steps` whole k-steps of this wave's pieces are still in flight (steps = 0, 1, 2: wave-uniform)
  auto wait_steps = [&](int steps) __attribute__((always_inline)) {
    if (steps == 0) wait_vm<0>();
    else if (steps == 1) { if (has_last) wait_vm<NJF + 1>(); else wait_vm<NJF>(); }
    else { if (has_last) wait_vm<2 * (NJF + 1)>(); else wait_vm<2 * NJF>(); }
  };

  // ---- fragment addresses ------------------------------------------------------------------------------------------
  // k-contiguous image: row R = r0 + l15, chunk (4 kb + lg) ^ (R & 7): byte = R 128 + ((lg ^ (l15 & 7)) << 4), kb = 1: ^ 64
  // row-contiguous image: see tr_frag
  const int nfA = (wm * TM * 16 + l15) * 128 + (((lg ^ (l15 & 7)) & 7) << 4);
  const int nfB = NPA * 1024 + (wn * TN * 16 + l15) * 128 + (((lg ^ (l15 & 7)) & 7) << 4);
  // row-contiguous image [64][BX]: the transposing read of k-rows 8 lg + tq (+ 4: second half of the fragment, + 32: second
  // half-step) takes 4 elements at column x + 4 tp; the chunk XOR is the same for k, k + 4 and k + 32 (tswz), so one per-lane
  // offset per fragment serves all four reads with immediate offsets
  const int tq = l15 >> 2, tp = l15 & 3;
  int tfA[TM], tfB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int col = wm * TM * 16 + i * 16 + 4 * tp, k = 8 * lg + tq;
    tfA[i] = k * (BM * 2) + (((col >> 3) ^ tswz<BM>(k)) << 4) + (col & 7) * 2;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * TN * 16 + j * 16 + 4 * tp, k = 8 * lg + tq;
    tfB[j] = NPA * 1024 + k * (BN * 2) + (((col >> 3) ^ tswz<BN>(k)) << 4) + (col & 7) * 2;
  }
  auto tr_frag = [&](const unsigned char* p, int pitch) __attribute__((always_inline)) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * pitch));
    const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, t);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragments of one 32-deep half of a k-step; `last`: the ragged last step zeroes the A slots beyond K
  // (`last` is a compile-time constant: the ragged step is peeled off the loop - as a run-time flag the zeroing became 56
  // v_cndmask per k-step in EVERY step, 40 % of the loop's vector instructions)
  auto load_frags = [&](const unsigned char* st, int kb, auto lastc, bf16x8 (&af)[TM], bf16x8 (&bfr)[TN]) __attribute__((always_inline)) {
    constexpr bool last = decltype(lastc)::value;
    const unsigned char* pa = st + (nfA ^ (kb << 6));
    const unsigned char* pb = st + (nfB ^ (kb << 6));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (TA) af[i] = tr_frag(st + tfA[i] + kb * 32 * (BM * 2), BM * 2);
      else af[i] = *reinterpret_cast<const bf16x8*>(pa + i * 2048);
    }
    if constexpr (last) if (kb * 32 + 8 * lg >= ktail) {          // this lane's 8 k-slots lie beyond K
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (TB) bfr[j] = tr_frag(st + tfB[j] + kb * 32 * (BN * 2), BN * 2);
      else bfr[j] = *reinterpret_cast<const bf16x8*>(pb + j * 2048);
    }
  };
  auto mma = [&](const bf16x8 (&af)[TM], const bf16x8 (&bfr)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  };
  // ONE barrier per k-step: before it every wave waits (counted vmcnt) for its own pieces of stage t; behind it stage t is
  // complete and stage t - 1 is free, so the pieces of step t + D are issued there and stay in flight under D steps of matrix
  // instructions.
  // Measured on M 3136 N 3072 K 3072 (round 4, us per launch, random operands; the library's kernel 53 - 55; vu_gemm.h's
  // 128 x 128 tile 126): THIS ORDER 73 - 81 (boxes differ).  Pieces of it, timed alone (results wrong by construction):
  //   matrix instructions only (fragments read once)            46   <- the floor of this tile on this chip: 224 CUs, and the
  //                                                                    clock an MFMA-dense loop holds on random data (~1.6 GHz)
  //   fragment reads only (160 KB of ds_read_b128 per step)     27.5 <- ~200 B / clock / CU: the LDS is a co-bottleneck
  //   reads + products, no DMA                                   63   (either hipcc's just-in-time reads or both halves up front)
  //   the DMA ring alone                                         34
  // and orders of the whole loop, all SLOWER than this one: pieces spread one by one between the rows of matrix instructions
  // 80 - 89; both halves' fragments requested before the first product 78 (hipcc sinks the reads back next to their uses), pinned
  // by sched_barrier 94, spread by sched_group_barrier 74 - 75 (no DMA); the barrier between the two halves with the next step's
  // first fragments in flight across it 95; LOAD / COMPUTE phases with the two waves of a SIMD one phase apart (an extra
  // barrier for half of the waves) 94; two independent 4-wave workgroups per CU (112 x 192) 76.  Reads and products add up
  // instead of overlapping (46 + 27.5 = 73.5): with eight waves of 112 x 48 the tile re-reads every A fragment four times
  // and every B fragment twice.  The library's shape - four waves, one per SIMD, 128 x 80 each (35 % fewer LDS bytes per
  // product), operands staged through registers - is the next step; it is a different kernel, not another order of this loop.
  constexpr int D = NST - 1;                  // k-steps of DMA in flight ahead of the one being multiplied
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nk) issue(d);
  bf16x8 a0[TM], b0[TN];
  auto kstep = [&](int t, auto lastc) __attribute__((always_inline)) {
    { const int ahead = nk - 1 - t; wait_steps(ahead < D - 1 ? ahead : D - 1); }     // stage t has landed (this wave's pieces); later steps may fly
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned char* st = smem + (t % NST) * STAGE;
    if (t + D < nk) issue(t + D);                                       // into the stage every wave finished reading before the barrier
    load_frags(st, 0, lastc, a0, b0);
    mma(a0, b0);
    load_frags(st, 1, lastc, a0, b0);
    mma(a0, b0);
  };
  for (int t = 0; t < nk - 1; ++t) kstep(t, std::false_type{});
  if (ktail != BK) kstep(nk - 1, std::true_type{});                       // ragged K: the slots beyond K multiply zeros
  else kstep(nk - 1, std::false_type{});
  // ---- epilogue: one half of the tile rows (one wm) per pass through an fp32 LDS tile -------------------------------------
  constexpr int LDC = BN + 4;
  constexpr int EP = (TM * 16 * LDC * 4 <= NST * STAGE) ? 1 : 2;      // sub-passes over a wave's row tiles when the whole half does not fit
  constexpr int IP = (TM + EP - 1) / EP, HM = IP * 16;
  static_assert(HM * LDC * 4 <= NST * STAGE, "C staging tile must fit the ring");
  float* Ct = reinterpret_cast<float*>(smem);
  const vu_rng rng = g.dropout ? vu_rng_resolve(g.rng) : g.rng;
#pragma unroll 1
  for (int pass = 0; pass < WM * EP; ++pass) {
    const int pw = pass / EP, i0 = (pass % EP) * IP;
    const int nrow = ((i0 + IP < TM ? i0 + IP : TM) - i0) * 16;       // rows of this pass
    asm volatile("" ::: "memory");
    __syncthreads();                                   // the ring (pass 0) / the previous pass's tile is no longer read
    if (wm == pw) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        if (i >= i0 && i < i0 + IP) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[((i - i0) * 16 + lg * 4 + r) * LDC + wn * TN * 16 + j * 16 + l15] = acc[i][j][r];
        }
    }
    __syncthreads();
    const int rbase = m0 + pw * TM * 16 + i0 * 16;
    if constexpr (CF) {
      constexpr int CPR = BN / 4;                      // float4 items per row
      for (int it = tid; it < nrow * CPR; it += NW * 64) {
        const int rl = it / CPR, c4 = (it - rl * CPR) * 4;
        const int row = rbase + rl, col = n0 + c4;
        if (row < g.M && col < g.N) {
          const float4 v = *reinterpret_cast<const float4*>(&Ct[rl * LDC + c4]);
          float* cp = reinterpret_cast<float*>(g.C) + (long long)row * g.ldc + col;
          float4 o = v;
          if (g.accumulate) { const float4 p = *reinterpret_cast<const float4*>(cp); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
          *reinterpret_cast<float4*>(cp) = o;
        }
      }
    } else {
      constexpr int CPR = BN / 8;                      // 16-byte bf16 items per row
      for (int it = tid; it < nrow * CPR; it += NW * 64) {
        const int rl = it / CPR, c8 = (it - rl * CPR) * 8;
        const int row = rbase + rl, col = n0 + c8;
        if (row < g.M && col < g.N) {
          const float4 a0 = *reinterpret_cast<const float4*>(&Ct[rl * LDC + c8]), a1 = *reinterpret_cast<const float4*>(&Ct[rl * LDC + c8 + 4]);
          float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          if (g.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(g.bias + col), b1 = *reinterpret_cast<const float4*>(g.bias + col + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          }
          const long long o = (long long)row * g.ldc + col;
          if (g.dropout) {
            const uint64_t idx = (uint64_t)row * (uint64_t)g.N + (uint64_t)col;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vu_keep(rng, idx + e) ? v[e] * rng.inv_keep : 0.f;
          }
          if (g.addend) {
            vu_f4 lo, hi;
            vu_ld8(g.addend + o, lo, hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += lo.v[e]; v[4 + e] += hi.v[e]; }
          }
          union U8 { uint4 u; bf16_t h[8]; } t;
#pragma unroll
          for (int e = 0; e < 8; ++e) t.h[e] = (bf16_t)v[e];
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + o) = t.u;
        }
      }
    }
  }
}

// colsum[n] += sum over the rows of in[row ld + n], one workgroup per 64 columns, every row, fixed order: the bias gradient
// that rides on a weight-gradient product must stay bit-reproducible (vu_k_colsum's row blocks end in float atomics)
__global__ __launch_bounds__(1024) void bg_colsum_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int rows, int ncols, long long ld) {
  __shared__ float red[128][65];
  const int v = threadIdx.x & 7, rl = threadIdx.x >> 3, c0 = blockIdx.x * 64 + v * 8;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < ncols)
    for (int r = rl; r < rows; r += 128) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(in + (long long)r * ld + c0);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += (float)x[i];
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][v * 8 + i] = a[i];
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < ncols) {
    float t = 0.f;
    for (int r = 0; r < 128; ++r) t += red[r][threadIdx.x];
    out[blockIdx.x * 64 + threadIdx.x] += t;
  }
}

inline int bg_mode() {        // VU_BGEMM: 0 = never (vu_gemm.h's tile), unset / 1 = where eligible
  static const int v = [] { const char* e = getenv("VU_BGEMM"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}

template <bool TA, bool TB, bool CF, int TM, int TN, int WM, int NST, int WN = 4>
int launch_tile(const bg_args& a0, hipStream_t st) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  bg_args a = a0;
  a.tiles_m = vu_cdiv(a.M, BM); a.tiles_n = vu_cdiv(a.N, BN);
  a.gm = 0; a.gn = 0;
  const int total = a.tiles_m * a.tiles_n;
  if (total % 8 == 0) {          // the most square split of the tile grid into 8 rectangles
    static const int cand[4][2] = {{2, 4}, {4, 2}, {8, 1}, {1, 8}};
    for (int c = 0; c < 4; ++c)
      if (a.tiles_m % cand[c][0] == 0 && a.tiles_n % cand[c][1] == 0) { a.gm = cand[c][0]; a.gn = cand[c][1]; break; }
  }
  constexpr size_t lds = (size_t)NST * ((BM + BN) / 8) * 1024;
  auto kern = bgemm_kernel<TA, TB, CF, TM, TN, WM, NST, WN>;
  static bool reserved = false;            // (per instantiation)
  if (!reserved) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vu_set_error("vu_bgemm: cannot reserve %zu bytes of LDS", lds);
      return VU_ELAUNCH;
    }
    reserved = true;
  }
  hipLaunchKernelGGL(kern, dim3(total), dim3(WM * WN * 64), lds, st, a);
  if (vu_prof_on()) {
    static const bool shapes = getenv("VU_PROF_SHAPES") != nullptr;
    char tag[112];
    const char* w4 = (WM * WN == 4 && BM == 224) ? ",4 waves" : "";
    if (shapes) snprintf(tag, sizeof(tag), "bgemm_kernel<%c%c,%s,%dx%d%s> M%d N%d K%d%s", TA ? 'T' : 'N', TB ? 'T' : 'N', CF ? "f32 acc" : "bf16", BM, BN, w4, a.M, a.N, a.K, a.dropout ? " +dropout" : "");
    else snprintf(tag, sizeof(tag), "bgemm_kernel<%c%c,%s,%dx%d%s>", TA ? 'T' : 'N', TB ? 'T' : 'N', CF ? "f32 acc" : "bf16", BM, BN, w4);
    vu_prof_note(tag, 2.0 * a.M * (double)a.N * a.K, 2.0 * ((double)a.M * a.K + (double)a.K * a.N) + (CF ? 8.0 : 2.0) * a.M * a.N);
  }
  return vu_check_launch("vu_gemm (bgemm)");
}

template <bool TA, bool TB, bool CF, int TM, int TN>
int launch(const bg_args& a, hipStream_t st) {
  // Tile choice.  224 x 192 (8 waves, 3 stages, one workgroup per CU) where it fills the chip (224 tiles at 64 images).  With
  // fewer than ~160 such tiles (16 - 32 images per GPU: 64 - 112) the k-contiguous-A forms take 112 x 192 (4 waves, 2 stages =
  // 78 KB: two independent workgroups per CU): measured M 1568 N 3072 K 3072 67 -> 52 us, M 784 63 -> 47 us; equal at 64 images
  // (76 us both).  VU_BGEMM_TILE = 0 / 1 / 2 forces 224 x 192 / 112 x 192 x 2 stages / 112 x 192 x 3 stages (A/B switch).
  // 16 images per GPU (M 784): 112 x 192 is 112 workgroups on 256 CUs; 112 x 128 (VU_BGEMM_TILE=3) makes 168 of two thirds the work.
  static const int force = [] { const char* e = getenv("VU_BGEMM_TILE"); return (e && e[0] >= '0' && e[0] <= '4') ? e[0] - '0' : -1; }();
  if constexpr (TN == 3) {
    if (force == 4) return launch_tile<TA, TB, CF, TM, 6, 2, 3, 2>(a, st);          // 224 x 192 by four waves of 112 x 96 (one per SIMD)
  }
  if constexpr (!TA) {       // (a row-contiguous A needs BM % 32 == 0)
    const long long big_tiles = (long long)vu_cdiv(a.M, 2 * TM * 16) * vu_cdiv(a.N, 4 * TN * 16);
    if constexpr (TN == 3) {
      const long long half_tiles = (long long)vu_cdiv(a.M, TM * 16) * vu_cdiv(a.N, 4 * TN * 16);
      // Up to 32 images per GPU: 112 x 64 tiles (44 KB of LDS: three workgroups per CU) while they fit ONE round of 768 slots, i.e. up
      // to 256 tiles of 112 x 192.  Round 6, in the step (same-box A/B, images/s): 16 images 2954 -> 3010 (Large 1592 -> 1620), 24 images
      // 3475 -> 3600, 32 images 4362 -> 4467; 64 images (448 such tiles, 1.75 rounds) 5736 -> 5604: not there.  Stand-alone with warm
      // operands the same launches measure EQUAL or slower (tools/gemm_small_batch.py: M 784 35.6 -> 32.7 us, M 1568 46.7 -> 47.6,
      // M 3136 N 768 10.7 -> 12.8) - in the step the operands come from HBM and three times the workgroups keep more loads in flight.
      // VU_BGEMM_T64 = 0 / 2 / 3: off / two / three stages; VU_BGEMM_T64_MAX: the tile-count limit
      static const int t64 = [] { const char* e = getenv("VU_BGEMM_T64"); return e ? atoi(e) : 2; }();
      static const int t64max = [] { const char* e = getenv("VU_BGEMM_T64_MAX"); return e ? atoi(e) : 257; }();
      if (t64 == 2 && half_tiles < t64max) return launch_tile<TA, TB, CF, TM, 1, 1, 2>(a, st);
      if (t64 == 3 && half_tiles < t64max) return launch_tile<TA, TB, CF, TM, 1, 1, 3>(a, st);
      // ... and 112 x 128 (60 KB: two per CU) while THOSE fit one round of 512 slots (40 - 48 images: 4437 -> 4539, 4958 -> 5013)
      if (force == 3 || (force < 0 && half_tiles * 3 <= 1024)) return launch_tile<TA, TB, CF, TM, 2, 1, 2>(a, st);
    }
    if (force == 1 || (force < 0 && big_tiles < 160)) return launch_tile<TA, TB, CF, TM, TN, 1, 2>(a, st);
    if (force == 2) return launch_tile<TA, TB, CF, TM, TN, 1, 3>(a, st);
  }
  return launch_tile<TA, TB, CF, TM, TN, 2, 3>(a, st);
}

}  // namespace

// 1 = done here, 0 = not eligible (the caller goes on with the general kernel), < 0 = error.  g in the ORIGINAL orientation.
int vu_bgemm_try(const vu_gemm_args& g, int c_float, hipStream_t st) {
  if (!bg_mode()) return 0;
  if (g.Z1 * g.Z2 != 1 || g.act != VU_ACT_NONE || g.aux || g.alpha != 1.f) return 0;
  if (!c_float && g.accumulate) return 0;
  if (c_float && (g.bias || g.dropout || g.addend)) return 0;
  // Round 5: also the 192-class Linear layers of level 2 (attention projection and its data gradient: N = K = 192, M = B * 784 rows).
  // As 128 x 128 tiles of vu_gemm.h they were 784 workgroups on 768 slots (a second, nearly empty round) with half of the second
  // column tile idle: 20.5 us stand-alone at 64 images; here 224 workgroups of one 224 x 192 tile, the three k-steps of the ring in
  // flight at once: 12.0 us (Base step at 64 images 11.78 -> 11.65 ms).  VU_BGEMM_SMALL=0 (read once) restores the tiled route.
  static const bool small_ok = [] { const char* e = getenv("VU_BGEMM_SMALL"); return !(e && e[0] == '0'); }();
  // ... and the short-K products with a wide output: the second feed-forward layer and the data gradient of the first at levels
  // 1 / 0 (K = 64 / 128, N = 768 / 3072; 21 - 23 us as vu_gemm.h tiles in the step, 9.5 - 10.6 stand-alone here: 11.595 -> 11.55 ms).
  // VU_BGEMM_SMALL=1 keeps them on the tiled route.
  static const bool shortk_ok = [] { const char* e = getenv("VU_BGEMM_SMALL"); return !(e && (e[0] == '0' || e[0] == '1')); }();
  static const int shortk_m = [] { const char* e = getenv("VU_BGEMM_SHORTK_M"); return e ? atoi(e) : 512; }();         // the smallest M and
  static const int shortk_mn = [] { const char* e = getenv("VU_BGEMM_SHORTK_MN"); return e ? atoi(e) : 21; }();      // log2(M N) of the short-K route (2048 / 22 until the 16-image batches were measured: Base 16 / GPU 2743 -> 2779, Large 16 / GPU 1476 -> 1486 images/s)
  const bool small = small_ok && !c_float && ((g.K == 192 && g.N == 192 && g.M >= 4096) ||
                                              (shortk_ok && g.K >= 64 && g.K < 512 && g.N >= 512 && g.M >= shortk_m && (long long)g.M * g.N >= (1ll << shortk_mn)));
  if (g.K % 8 != 0 || g.N % 8 != 0 || g.ldc % 8 != 0) return 0;
  if (!small) {
    if (g.K < 512 || g.N < 512 || g.M < 512) return 0;
    static const int min_work = [] { const char* e = getenv("VU_BGEMM_MINWORK"); return e ? atoi(e) : 30; }();      // log2 of M N K below which the tiled route keeps the product (round 5: 32 -> 30, the 768-class layers at 16 - 32 images: Base 16 / GPU 2699 -> 2728, Large 16 1445 -> 1467, Base 32 4160 -> 4194 images/s)
    if (c_float ? ((long long)g.M * g.N < (4ll << 20)) : ((long long)g.M * g.N * g.K < (1ll << min_work))) return 0;
  }
  if (!((g.sAk == 1) != (g.sAm == 1)) || !((g.sBn == 1) != (g.sBk == 1))) return 0;
  const bool TA = g.sAm == 1, TB = g.sBn == 1;
  if (TA && !TB) return 0;                                 // (no caller has this form)
  if (TA && g.M % 8 != 0) return 0;
  if (g.colsum && !(c_float && ((g.colsum_side == 1 && TA) || (g.colsum_side == 2 && TB)))) return 0;   // (bg_colsum_kernel reads the row-contiguous operand)
  bg_args a;
  a.A = (const bf16_t*)g.A; a.B = (const bf16_t*)g.B; a.C = g.C; a.M = g.M; a.N = g.N; a.K = g.K;
  a.lda = TA ? g.sAk : g.sAm; a.ldb = TB ? g.sBk : g.sBn; a.ldc = g.ldc;
  if (a.lda % 8 != 0 || a.ldb % 8 != 0 || a.lda < (TA ? g.M : g.K) || a.ldb < (TB ? g.N : g.K) || g.ldc < g.N) return 0;
  if (((uintptr_t)g.A | (uintptr_t)g.B | (uintptr_t)g.C | (uintptr_t)g.addend | (uintptr_t)g.bias) & 15) return 0;
  if ((long long)a.lda * (TA ? g.K : g.M) * 2 >= (1ll << 32) || (long long)a.ldb * (TB ? g.K : g.N) * 2 >= (1ll << 32)) return 0;   // 32-bit DMA offsets
  a.bias = g.bias; a.addend = (const bf16_t*)g.addend; a.dropout = g.dropout; a.rng = g.rng; a.accumulate = g.accumulate;
  a.tiles_m = a.tiles_n = a.gm = a.gn = 0;
  int rc;
  if (c_float) {
    if (TA && TB) rc = launch<true, true, true, 7, 3>(a, st);
    else if (!TA && !TB) rc = launch<false, false, true, 7, 3>(a, st);
    else rc = launch<false, true, true, 7, 3>(a, st);
  } else {
    if (TA && TB) rc = launch<true, true, false, 7, 3>(a, st);
    else if (!TA && !TB) rc = launch<false, false, false, 7, 3>(a, st);
    else rc = launch<false, true, false, 7, 3>(a, st);
  }
  if (rc < 0) return rc;
  if (g.colsum) {           // bias gradient riding on a weight-gradient product: column sums over k of A (side 1) or B (side 2)
    const bool a_side = g.colsum_side == 1;
    const int nc = a_side ? g.M : g.N;
    hipLaunchKernelGGL(bg_colsum_kernel, dim3((unsigned)((nc + 63) / 64)), dim3(1024), 0, st, (const bf16_t*)(a_side ? g.A : g.B), g.colsum, g.K, nc, a_side ? g.sAk : g.sBk);
    if (vu_prof_on()) vu_prof_note("bg_colsum_kernel", 0.0, (double)g.K * nc * 2.0);
    rc = vu_check_launch("vu_gemm (bgemm: bias-gradient sums)");
    if (rc < 0) return rc;
  }
  return 1;
}
