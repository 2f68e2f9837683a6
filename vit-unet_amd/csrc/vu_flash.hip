// Non-materialising re-attention (model.py:155-161) for long rows and small head dims: bf16 storage, N % 16 == 0,
// 8 heads of 8 / 24 / 32 features (Base / Large level 2: N = 784, d = 24; 512x512 inputs: d = 8 / 32) or 4 heads of 16 / 32 / 48
// (Lite: N = 3136 with d = 12 zero-padded to 16 by the caller, N = 784 with d = 48 as two k-steps per logits product).
//
// The (B,h,N,N) attention maps never exist in HBM.  BatchNorm over the head-mixed maps needs batch statistics of
// A_g = sum_h W[g,h] P~_h + c_g BEFORE the PV product (SURVEY 7 hard part 1), so the forward is two recompute passes:
//   stats pass   per 16-query tile: QK^T on MFMA for ALL heads, row max / row sum (two sweeps over the keys) ->
//                lse2[b,h,i] (log2 domain); a third sweep recomputes P~ = dropout(P) and accumulates the first moments
//                sum (P~_h - 1/N) and the h x h cross moments sum (P~_h - 1/N)(P~_h' - 1/N); var(A_g) = W_g^T Cov W_g
//                in the fp64 finalize.  No map write.
//   apply pass   recompute P~ from lse2, mix the heads in registers with the folded table (gamma rstd W / keep, folded
//                bias), round A^ to bf16 as the B operand of O^T = V^T A^^T.  No map read.
// and the backward is recompute sweeps of the same tile body (row term delta_h = sum_j P dP first, then dq; dk / dv in a
// key-major sweep), with the BatchNorm-backward means taken from dO, O, v (vu_k_bn_bwd_small).
//
// Tile body (all sweeps): the SWAPPED product S^T = K Q^T with v_mfma_f32_16x16x32_bf16: accumulator row = key
// 4 (lane >> 4) + r, column = query (lane & 15).  Since d <= 32 one MFMA is the whole k-loop of a (head, tile), so a wave
// holds the 16 x 16 tile of ALL heads in 4 H registers and the 8 x 8 head mix is lane-local VALU work in fp32
// (exactly the reference's arithmetic; no bf16 rounding of the probabilities at all).  Reductions over keys are
// in-lane sums plus two shuffles; products that contract over keys (PV, dq) take the accumulators of two tiles as the
// B operand with no lane movement, their A operands come out of the row-major K / V chunk through ds_read_b64_tr_b16.
// Algorithmic HBM traffic: q, k, v, O (+ dO, dq, dk, dv) only: ~10 B N D per module instead of 11 E.
#include <stdio.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include "vu_kernels.h"
#include "vu_flash.h"

#define VU_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != VU_OK) return _rc; \
  } while (0)

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// waves per SIMD the 4-head backward sweeps are compiled for (A/B: -DVU_FLASH_V1_W4=2 restores the 8-head value)
#ifndef VU_FLASH_V1_W4
#define VU_FLASH_V1_W4 4
#endif
#define VU_FLASH_V1_WAVES(H, DH) (((H) == 4 && (DH) <= 32) ? VU_FLASH_V1_W4 : 2)      // (d = 48: LDS allows two workgroups per CU anyway)
// Round 5, the 4-head sweeps' synchronous chunk staging (load_chunk between two barriers; at four waves per SIMD they have no
// registers to prefetch into).  Ablated (VU_V1_NOSTAGE=1, results wrong) Lite at 32 images goes 15.70 -> 14.85 ms: delta + dq -16 %,
// row statistics + moments -10 %, apply -16 %.  Built and measured: double-buffered LDS-DMA staging as in flash3_dk_kernel (generic
// padded-image DMA for any (H, d), row constants of the key-major sweep by 4-byte DMA), HALF the tiles per buffer so that the LDS
// and the four workgroups per CU stay: correct (the 4-head parity tests pass) and NO faster - 15.70 ms; delta + dq -2 %, apply -3 %,
// dk +5 % - because what the ablation removed was the barrier pair per chunk (twice as frequent with half-size buffers), not the
// load latency, which four to five waves per SIMD already hide.  Not kept.
#ifndef VU_V1_NOSTAGE
#define VU_V1_NOSTAGE 0          // measurement builds (results wrong): the 4-head sweeps stage only their first chunk
#endif
template <int H, int DH> struct FC {
  static constexpr int D = H * DH;
  // LDS row pitch of a (token x feature) chunk, elements: 16 B of (zeroed) pad.  (Rows 400 B apart at d = 24 start 4 banks apart, so
  // the four rows x 32 B a 16-lane group of a transposing read touches overlap pairwise - SQ_LDS_BANK_CONFLICT is 31 - 39 % of the
  // sweeps' LDS cycles; a 32-byte pad puts them 8 banks apart and changed NOTHING measurable at 64 images (round 5: dk 255 -> 250 us,
  // every other sweep within 1 %): the sweeps do not wait for LDS bandwidth.)
  static constexpr int PITCH = D + 8;
  static constexpr int KS = DH / 8;              // valid 8-feature k-slots of the QK^T product (NK MFMAs of k = 32 each)
  static constexpr int NK = (DH + 31) / 32;      // k-steps of a logits-shaped product (1 for d <= 32; d = 48: 32 + 16)
  static constexpr int DT = (DH + 15) / 16;      // 16-row tiles of the head dim in the transposed products
  static constexpr int VPR = D / 8;              // 16-byte vectors per row
  static constexpr int NMOM = H + H * (H + 1) / 2;
};

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// One row of an H x H coefficient table in LDS (all lanes read the same address: broadcast).  The mixing loops fetch
// row i+1 before the arithmetic of row i (with only two waves per SIMD a load that is waited for at once costs its
// whole latency): `LDS_FENCE` keeps the compiler from either hoisting every row out of the tile loop (64 + 64 extra
// registers = the second wave per SIMD) or sinking the prefetch back next to its use.
template <int H> struct RowW { float w[H]; };
template <int H>
__device__ __forceinline__ RowW<H> ld_row(const float* p) {
  RowW<H> r;
#pragma unroll
  for (int h4 = 0; h4 < H; h4 += 4) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(p + h4);
    r.w[h4] = w4[0]; r.w[h4 + 1] = w4[1]; r.w[h4 + 2] = w4[2]; r.w[h4 + 3] = w4[3];
  }
  return r;
}
#define LDS_FENCE() asm volatile("" ::: "memory")
// The transposed-mix table FWkT of the 4-head backward sweeps lives in REGISTERS (16 per lane), not in the LDS struct: with its rows
// prefetched from LDS inside the head loop (rounds 2 - 3) the delta / dq / dk sweeps lost bit-reproducibility whenever ANOTHER
// PROCESS computed on the same GPU - delta (and with it dq, dk) of 1 - 2 % of the 16-query tiles off in the last bits, every other
// buffer of the module identical (tools/attn_ws_diff.py).  Same arithmetic, same order; -DVU_V1_FW_LDS=1 restores the LDS
// rows for the A/B record.  (8 heads: 64 registers would not fit; that instantiation is not used - the 8-head form is flash2_*.)
#if defined(VU_V1_FW_LDS) && VU_V1_FW_LDS
#define VU_V1_FW_DECL ((void)0)
#define VU_V1_FW_ROW(i) ld_row<H>(tb->FWkT + (i) * H)
#else
#define VU_V1_FW_DECL \
  RowW<H> fwr[H <= 4 ? H : 1]; \
  if constexpr (H <= 4) { \
    _Pragma("unroll") for (int hh_ = 0; hh_ < H; ++hh_) { _Pragma("unroll") for (int gg_ = 0; gg_ < H; ++gg_) fwr[hh_].w[gg_] = stats[VU_BN_STATS_FWK(H) + gg_ * H + hh_]; } \
  }
#define VU_V1_FW_ROW(i) (H <= 4 ? fwr[H <= 4 ? (i) : 0] : ld_row<H>(tb->FWkT + (i) * H))
#endif

// rows [0, nrows) of a row-major (., D) bf16 matrix -> LDS chunk with pitch PITCH.  Four independent 16-byte loads in
// flight per thread (named registers and clamped unconditional loads: an indexed array or a load-or-skip select ends up
// in scratch memory / serialized loads with hipcc).
template <int H, int DH>
__device__ __forceinline__ void load_chunk(bf16_t* dst, const bf16_t* __restrict__ src, int nrows, int tid, int nthr) {
  typedef FC<H, DH> C;
  const int total = nrows * C::VPR;
  auto src_of = [&](int v) { const int vv = v < total ? v : total - 1; const int r = vv / C::VPR, c = vv - r * C::VPR; return src + (long long)r * C::D + c * 8; };
  auto put = [&](int v, const uint4& x) { if (v < total) { const int r = v / C::VPR, c = v - r * C::VPR; *reinterpret_cast<uint4*>(dst + r * C::PITCH + c * 8) = x; } };
  for (int v0 = tid; v0 < total; v0 += 4 * nthr) {
    const uint4 x0 = *reinterpret_cast<const uint4*>(src_of(v0));
    const uint4 x1 = *reinterpret_cast<const uint4*>(src_of(v0 + nthr));
    const uint4 x2 = *reinterpret_cast<const uint4*>(src_of(v0 + 2 * nthr));
    const uint4 x3 = *reinterpret_cast<const uint4*>(src_of(v0 + 3 * nthr));
    put(v0, x0); put(v0 + nthr, x1); put(v0 + 2 * nthr, x2); put(v0 + 3 * nthr, x3);
  }
}
template <int H, int DH>
__device__ __forceinline__ void zero_pads(bf16_t* dst, int rows, int tid, int nthr) {
  typedef FC<H, DH> C;
  for (int r = tid; r < rows; r += nthr) *reinterpret_cast<uint4*>(dst + r * C::PITCH + C::D) = make_uint4(0, 0, 0, 0);
}

// Register-staged chunk for software prefetch: fetch() issues the global loads of the NEXT chunk before the tile loop of
// the current one, commit() writes them to LDS after the barrier that ends it - the HBM / L2 latency of a chunk (4 dependent
// round trips, ~20 % of a sweep with the load-then-compute form: rocprofv3 SQ_WAIT_ANY, profiles/r02b) hides under the
// arithmetic.  ROWS rows of D elements by NTHR threads: NV 16-byte vectors per thread.
template <int H, int DH, int ROWS, int NTHR>
struct ChunkStage {
  typedef FC<H, DH> C;
  static constexpr int TOTAL = ROWS * C::VPR;
  static constexpr int NV = (TOTAL + NTHR - 1) / NTHR;
  static_assert(NV <= 12, "ChunkStage: more than 12 vectors per thread");
  uint4 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11;      // named registers (an indexed array ends up in scratch memory)
  __device__ __forceinline__ static uint4 ld(const bf16_t* __restrict__ src, int total, int x) {
    const int xx = x < total ? x : total - 1;                   // clamped: an unconditional load (commit() drops it)
    const int r = xx / C::VPR, c = xx - r * C::VPR;
    return *reinterpret_cast<const uint4*>(src + (long long)r * C::D + c * 8);
  }
  __device__ __forceinline__ static void st(bf16_t* dst, int total, int x, const uint4& val) {
    if (x < total) { const int r = x / C::VPR, c = x - r * C::VPR; *reinterpret_cast<uint4*>(dst + r * C::PITCH + c * 8) = val; }
  }
#define VU_STAGE_EACH(OP) \
  if constexpr (NV > 0) { OP(0, v0) } if constexpr (NV > 1) { OP(1, v1) } if constexpr (NV > 2) { OP(2, v2) } \
  if constexpr (NV > 3) { OP(3, v3) } if constexpr (NV > 4) { OP(4, v4) } if constexpr (NV > 5) { OP(5, v5) } \
  if constexpr (NV > 6) { OP(6, v6) } if constexpr (NV > 7) { OP(7, v7) } if constexpr (NV > 8) { OP(8, v8) } \
  if constexpr (NV > 9) { OP(9, v9) } if constexpr (NV > 10) { OP(10, v10) } if constexpr (NV > 11) { OP(11, v11) }
  __device__ __forceinline__ void fetch(const bf16_t* __restrict__ src, int nrows, int tid) {
    const int total = nrows * C::VPR;
#define VU_STAGE_LD(i, reg) reg = ld(src, total, tid + (i) * NTHR);
    VU_STAGE_EACH(VU_STAGE_LD)
#undef VU_STAGE_LD
  }
  __device__ __forceinline__ void commit(bf16_t* dst, int nrows, int tid) const {
    const int total = nrows * C::VPR;
#define VU_STAGE_ST(i, reg) st(dst, total, tid + (i) * NTHR, reg);
    VU_STAGE_EACH(VU_STAGE_ST)
#undef VU_STAGE_ST
  }
#undef VU_STAGE_EACH
};

// B operand of the logits product for one 16-token tile held in registers: lane (token l15, k-slot g4) has features
// h DH + 8 g4 .. + 7 of its token, zero where 8 g4 >= DH (so whatever the other operand holds there is multiplied by 0).
// (f[h NK + kh]: k-step kh of head h)
template <int H, int DH, int NF>
__device__ __forceinline__ void load_stationary(bf16x8 (&f)[NF], const bf16_t* __restrict__ rowp, int g4) {
  typedef FC<H, DH> C;
  static_assert(NF == H * C::NK, "one fragment per (head, k-step)");
  if constexpr (C::NK == 1) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      if (g4 < C::KS) f[h] = *reinterpret_cast<const bf16x8*>(rowp + h * DH + 8 * g4);
      else f[h] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int kh = 0; kh < C::NK; ++kh) {
      if (4 * kh + g4 < C::KS) f[h * C::NK + kh] = *reinterpret_cast<const bf16x8*>(rowp + h * DH + 32 * kh + 8 * g4);
      else f[h * C::NK + kh] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
}

// S^T tile of every head: A = chunk rows (tokens 16 kc + l15 of the chunk), B = stationary fragments.
template <int H, int DH, int NF>
__device__ __forceinline__ void tile_logits(f32x4 (&acc)[H], const bf16_t* Kc, int kc, const bf16x8 (&qf)[NF], int l15, int g4) {
  typedef FC<H, DH> C;
  static_assert(NF == H * C::NK, "one fragment per (head, k-step)");
  if constexpr (C::NK == 1) {      // d <= 32: one MFMA is the whole k-loop (kept verbatim: the 8-head sweeps sit at their register limit)
    const int gk = g4 < C::KS ? g4 : C::KS - 1;                 // slots >= KS meet zeros: re-read a valid slot (finite data)
    const bf16_t* krow1 = Kc + (kc * 16 + l15) * C::PITCH + 8 * gk;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow1 + h * DH);
      acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[h], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    return;
  }
  const bf16_t* krow = Kc + (kc * 16 + l15) * C::PITCH;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < C::NK; ++kh) {
      const int ks = C::KS - 4 * kh < 4 ? C::KS - 4 * kh : 4;   // valid slots of this k-step (compile time after unrolling)
      const int gk = g4 < ks ? g4 : ks - 1;                     // slots >= ks meet zeros: re-read a valid slot (finite data)
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + h * DH + 32 * kh + 8 * gk);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[h * C::NK + kh], a, 0, 0, 0);
    }
    acc[h] = a;
  }
}

// transposed A operand of v_mfma_f32_16x16x16_bf16 (rows = 16 features f0 .. f0+15 of the chunk, k = 16 tokens): k-slot
// 4 g4 + j <- token row0 + 4 g4 + j, which is the order of ONE accumulator tile used as the B operand (register r = key
// 4 g4 + r): the products that contract over keys take a logits-shaped tile as it stands, tile by tile.
template <int PITCH>
__device__ __forceinline__ s16x4 tr_operand(const bf16_t* Xc, int row0, int f0, int l15, int g4) {
  const int qq = l15 >> 2, pp = l15 & 3;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Xc + (row0 + 4 * g4 + qq) * PITCH + f0 + 4 * pp));
}
__device__ __forceinline__ f32x4 mfma16(const s16x4& a, const s16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// two floats -> one register of two bf16 (round to nearest even): one v_cvt_pk_bf16_f32.  (Packing element by element
// through bf16 vectors made hipcc emit a convert per element plus v_perm / v_mov shuffles: 2.5x the instructions.)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ unsigned pk2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// acc + (the two bf16 halves of w): v_dot2c_f32_bf16 against (1, 1) - the sum of bf16-ROUNDED values exactly as an MFMA sees
// them, one instruction per packed pair
__device__ __forceinline__ float sum2_bf16(unsigned w, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, 0x3F803F80u), acc, false);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ s16x4 pack4s(const f32x4& a) {
  const u32x2_t t = {pk2(a[0], a[1]), pk2(a[2], a[3])};
  return __builtin_bit_cast(s16x4, t);
}
// a - (float)bf16(a), element-wise, given the packed bf16 of a
__device__ __forceinline__ f32x4 residual4(const f32x4& a, const s16x4& packed) {
  const u32x2_t t = __builtin_bit_cast(u32x2_t, packed);
  return f32x4{a[0] - bf_lo(t[0]), a[1] - bf_hi(t[0]), a[2] - bf_lo(t[1]), a[3] - bf_hi(t[1])};
}

__device__ __forceinline__ bf16x4 pack4(const f32x4& a) {
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  const f32x2 lo = {a[0], a[1]}, hi = {a[2], a[3]};
  const u2 t = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, b2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, b2))};
  return __builtin_bit_cast(bf16x4, t);
}
__device__ __forceinline__ bf16x8 join8(const bf16x4& a, const bf16x4& b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// Dropout mask of the attention maps in this form ("quad" scheme; the materialised kernels keep vu_keep's pair scheme,
// the oracle replays both): ONE 32-bit hash word serves the 4 consecutive keys 4 jg .. 4 jg + 3 of a map row, 8 bits each:
// key r is kept when byte r >= thr8 = round(256 p), so the drop probability is thr8 / 256 and 1 / keep = 256 / (256 - thr8)
// (vu_flash_quad_rng).  Every pass of the forward and the backward recomputes the word, so the hash is built from
// full-rate instructions only: two rounds of (24-bit multiply-add, xor-shift) - v_mul_lo_u32 runs at a quarter of the
// rate.  Word index x = (sample * N + query row) * (N / 4) + jg (32 bits: B N N < 2^34), shared by the heads (vu_quad_head).  Measured on 2^20 consecutive words, three
// keys: keep rate 0.8008 +- 0.0005 per byte lane, every pairwise correlation tested (byte lanes, neighbouring words, rows,
// heads) below 0.006, kept-per-row variance 0.98 - 1.03 of binomial (tests/test_oracle_golden.py::test_quad_mask_statistics).
__device__ __forceinline__ uint32_t vu_quad_word(uint32_t x, uint32_t k0, uint32_t k1) {
  const uint32_t a = x ^ k0;
  uint32_t y = __umul24(a, 0xB5297Bu) + (a >> 12);       // v_mad_u32_u24: low 24 bits of a times the constant
  y ^= y >> 16;
  y = __umul24(y, 0x9E3779u) + (y >> 12);
  y ^= y >> 16;
  return y + k1;
}
// The 8 heads of one (query row, key quad) share ONE two-round word (`base`, indexed without the head); head h takes a
// third round with its own multiplier: 3 instructions per head instead of 9 (the hash was 15 % of a sweep's VALU work).
// Measured on 2 M words (oracle, test_quad_mask_statistics): keep rate 0.8007 - 0.8010 per head (target 0.8008), largest
// correlation between two heads' masks 8e-4, between byte lanes of different heads 2e-3.
__device__ __forceinline__ uint32_t vu_quad_head(uint32_t base, uint32_t base_sh, int h) {
  // (odd, below 2^23: with bit 23 set hipcc emits the quarter-rate v_mad_u64_u32 instead of v_mad_u32_u24)
  constexpr uint32_t C[8] = {0x1E3779u, 0x35297Bu, 0x68E31Du, 0x7FEB35u, 0x42B2AFu, 0x65EBCBu, 0x27D4EBu, 0x165667u};
  const uint32_t y = __umul24(base, C[h & 7]) + base_sh;
  return y ^ (y >> 16);
}
struct keep4_t { uint32_t w, thr; };
__device__ __forceinline__ keep4_t keep4(const vu_rng& rng, uint32_t x) {
  keep4_t kp = {0xffffffffu, rng.thr};
  if (rng.thr) kp.w = vu_quad_word(x, rng.k0, rng.k1);
  return kp;
}
__device__ __forceinline__ bool kept(const keep4_t& kp, int r) { return ((kp.w >> (8 * r)) & 0xffu) >= kp.thr; }

// XCD-aware work order: the workgroups of one sample stream the same K / V from L2, so they are dealt to one XCD
// (blocks id and id + 8 share an XCD) when the sample count is a multiple of 8.  Speed only.
__device__ __forceinline__ void work_item(int id, int B, int per, int& b, int& g) {
  if ((B & 7) == 0) { const int x = id & 7, r = id >> 3; b = x + 8 * (r / per); g = r % per; }
  else { b = id / per; g = id % per; }
}

// =============================================================================================
// stats pass
// =============================================================================================
template <int H, int DH, int WPB, int CK>
__global__ __launch_bounds__(WPB * 64) void flash_stats_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                               float* __restrict__ lse2, float* __restrict__ rinv,
                                                               float* __restrict__ partials, float* __restrict__ pk_out,
                                                               float* __restrict__ rinv_b, int B, int N,
                                                               float c, vu_rng rng_in, int want_moments) {
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);                       // [CK*16][PITCH]
  float* red = reinterpret_cast<float*>(Kc + CK * 16 * C::PITCH);        // [WPB][NMOM]
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int ntiles = N >> 4, per = (ntiles + WPB - 1) / WPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * WPB + wave;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  bf16x8 qf[H * C::NK];
  load_stationary<H, DH>(qf, q + ((long long)b * N + qrow) * C::D, g4);
  zero_pads<H, DH>(Kc, CK * 16, tid, WPB * 64);
  const int nchunks = (ntiles + CK - 1) / CK;

  float mx[H], sm[H];
#pragma unroll
  for (int h = 0; h < H; ++h) { mx[h] = -3.0e38f; sm[h] = 0.f; }
  // sweep 1 (round 4: ONE pass instead of a maxima pass and a sums pass - the recurrence of flash_rowstats_kernel): every lane
  // carries a running (max, sum) pair per head over its own keys in the log2 domain and rescales the sum when its maximum moves;
  // the four lane groups of a query are merged at the end (every lane ends up with the row's pair)
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    if (!(VU_V1_NOSTAGE && ch > 0)) {
    __syncthreads();
    load_chunk<H, DH>(Kc, kb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    __syncthreads();
    }
    if (active)
      for (int kc = 0; kc < nt; ++kc) {
        f32x4 acc[H];
        tile_logits<H, DH>(acc, Kc, kc, qf, l15, g4);
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float tm = fmaxf(fmaxf(acc[h][0], acc[h][1]), fmaxf(acc[h][2], acc[h][3])) * c;      // c > 0
          const float mn = fmaxf(mx[h], tm);
          float a = sm[h] * fexp2(mx[h] - mn);
#pragma unroll
          for (int r = 0; r < 4; ++r) a += fexp2(fmaf(acc[h][r], c, -mn));
          sm[h] = a;
          mx[h] = mn;
        }
      }
  }
  float lse[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float m = mx[h], s_ = sm[h];
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s_, o, 64);
      const float mn = fmaxf(m, m2);
      s_ = s_ * fexp2(m - mn) + s2 * fexp2(m2 - mn);
      m = mn;
    }
    lse[h] = m + log2f(s_);
    if (active && g4 == 0) {
      lse2[((long long)b * H + h) * N + qrow] = lse[h];
      rinv[((long long)b * H + h) * N + qrow] = 1.0f / (s_ * fexp2(m - lse[h]));      // (row_norm_note below)
    }
  }
  if (!want_moments) return;                                      // (uniform: eval mode needs the row statistics only)

  // sweep 3: first and cross moments of the centred dropped probabilities u_h = P~_h - 1/N
  float s1[H], s2[H * (H + 1) / 2];
#pragma unroll
  for (int h = 0; h < H; ++h) s1[h] = 0.f;
#pragma unroll
  for (int i = 0; i < H * (H + 1) / 2; ++i) s2[i] = 0.f;
  // pk_out (round 4): V = sum_k bf16(P) k per head and s_b = sum_k bf16(P) (row_norm_note) ride on this sweep, as in the 8-head
  // form's moments sweep, so that the backward forms dq in its delta sweep (dq = scale (U - delta~ V), flash_bwd_delta_kernel<DQ>)
  const bool want_pk = pk_out != nullptr;          // (kernel-uniform)
  f32x4 pacc[H][C::DT];
  float sb[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    sb[h] = 0.f;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) pacc[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float cen = 1.0f / (float)N;
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);          // quad words per (sample, head) map
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    if (!(VU_V1_NOSTAGE && ch > 0)) {
    __syncthreads();
    load_chunk<H, DH>(Kc, kb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    __syncthreads();
    }
    if (active)
      for (int kc = 0; kc < nt; ++kc) {
        f32x4 acc[H];
        tile_logits<H, DH>(acc, Kc, kc, qf, l15, g4);
        const uint32_t wt = wlane + 4u * (uint32_t)(ch * CK + kc);
        const uint32_t base = keep4(rng, wt).w, base_sh = base >> 12;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          keep4_t kp;
          kp.thr = rng.thr;
          kp.w = rng.thr ? vu_quad_head(base, base_sh, h) : 0u;
          f32x4 pu;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fexp2(fmaf(acc[h][r], c, -lse[h]));
            pu[r] = p;
            acc[h][r] = kept(kp, r) ? fmaf(p, rng.inv_keep, -cen) : -cen;
          }
          if (want_pk) {           // a logits-shaped tile is the B operand of a key-contracting product as it stands
            const s16x4 bop = pack4s(pu);
            const u32x2_t bw = __builtin_bit_cast(u32x2_t, bop);
            sb[h] = sum2_bf16(bw[1], sum2_bf16(bw[0], sb[h]));
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
              pacc[h][dt] = mfma16(tr_operand<C::PITCH>(Kc, kc * 16, h * DH + 16 * dt, l15, g4), bop, pacc[h][dt]);
          }
        }
        int idx = 0;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          s1[h] += (acc[h][0] + acc[h][1]) + (acc[h][2] + acc[h][3]);
#pragma unroll
          for (int h2 = 0; h2 <= h; ++h2, ++idx)
#pragma unroll
            for (int r = 0; r < 4; ++r) s2[idx] = fmaf(acc[h][r], acc[h2][r], s2[idx]);
        }
      }
  }
  if (want_pk) {
    float* prow = pk_out + ((long long)b * N + qrow) * C::D;
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int f = 16 * dt + 4 * g4;              // accumulator row = head feature, column = query l15
        if (active && f < DH) *reinterpret_cast<f32x4*>(prow + h * DH + f) = pacc[h][dt];
      }
      float x = sb[h];                                // the four lane groups of a query hold its four key quarters
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      // rinv keeps the fp32 form (the dk sweep multiplies fp32 probabilities: its delta must be divided by THEIR row sum,
      // section 2 "saturated rows" point 2); 1 / sum_k bf16(P) - the divisor of delta~ in the fused dq form - goes to rinv_b
      if (active && g4 == 0) rinv_b[((long long)b * H + h) * N + qrow] = 1.0f / x;
    }
  }
  // wave -> workgroup -> one partial row per workgroup
  __syncthreads();
#pragma unroll
  for (int h = 0; h < H; ++h) { const float v = vu_wave_sum(active ? s1[h] : 0.f); if (lane == 0) red[wave * C::NMOM + h] = v; }
#pragma unroll
  for (int i = 0; i < H * (H + 1) / 2; ++i) { const float v = vu_wave_sum(active ? s2[i] : 0.f); if (lane == 0) red[wave * C::NMOM + H + i] = v; }
  __syncthreads();
  for (int i = tid; i < C::NMOM; i += WPB * 64) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) a += red[w * C::NMOM + i];
    partials[(long long)blockIdx.x * C::NMOM + i] = a;
  }
}

// Row statistics alone, ONE sweep (the v2 forward's first pass): every lane carries a running (max, sum) pair per head
// over its own keys (4 per tile) in the log2 domain and rescales the sum when its maximum moves; the four lane groups of a
// query are merged at the end.  The next K chunk is fetched into registers while the current one is consumed.

// ---- split of the streamed axis over the waves of a workgroup (KS = 2) ------------------------------------------------
// At small batches the sweeps launch fewer workgroups than CUs (Base level 2 at 16 images: 208) and every SIMD runs ONE
// wave of a latency-bound chain.  With KS = 2 a workgroup owns two tiles instead of four: waves 2p and 2p + 1 share own
// tile p and take the even / odd tiles of the streamed operand (the chunk staging stays cooperative and unchanged), so the
// grid doubles and every SIMD gets two waves.  At the end the odd wave parks its accumulators in LDS (the operand images are
// dead by then), the even wave adds them and runs the epilogue alone.  Sums are re-associated (even tiles + odd tiles), so
// results differ from KS = 1 by rounding; a forward and its backward always run with the same split only by convention of
// the launcher - nothing depends on it (lse, rinv, pk are stored values).
template <int NV>
__device__ __forceinline__ void pair_park(const f32x4 (&v)[NV], float* scratch, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i) reinterpret_cast<f32x4*>(scratch)[i * 64 + lane] = v[i];
}
template <int NV>
__device__ __forceinline__ void pair_take(f32x4 (&v)[NV], const float* scratch, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const f32x4 x = reinterpret_cast<const f32x4*>(scratch)[i * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[i][r] += x[r];
  }
}

template <int H, int DH, int WPB, int CK, int KS = 1>
// row_norm_note: every later sweep recomputes P = exp2(c s - lse) from the ONE stored float lse = m + log2(sum).  When the
// logits are huge (un-normalised skip outputs through saturating e4m3 operands reach c s ~ 1.6e6, where a float resolves
// 0.125) the rounding of that sum leaves every P of the row off by one common factor g = sum 2^(m - lse) (up to 4.4 % at
// 2^21; below 0.1 % for |c s| < 2^15): harmless as a factor, but the softmax backward dS = P~ dP~ - P delta relies on
// sum_k P = 1 to cancel, and with sum_k P = g it leaves a spurious - (g - 1) P delta - for saturated (one-hot) rows, whose
// true dS is 0, that term IS the gradient (measured: 100 - 400 % error of the q / k convolution weight gradients of the
// 512 x 512 configuration's first level-1 decoder block).  rinv = 1 / g is stored beside lse; the backward uses delta / g,
// which restores the cancellation exactly (dS becomes g times the true dS: the common factor again).
// (measured, round 3: capped at 128 registers for four waves per SIMD - 832 work groups in ONE round - it is slower, 103 vs
// 96 us: the sweep is bound by its exp2 chain per wave, not by the second, nearly empty round)
__global__ __launch_bounds__(WPB * 64, 2) void flash_rowstats_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                      float* __restrict__ lse2, float* __restrict__ rinv, int B, int N, float c) {
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  constexpr int TPB = WPB / KS;
  const int ntiles = N >> 4, per = (ntiles + TPB - 1) / TPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * TPB + wave / KS, ksh = wave % KS;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  bf16x8 qf[H * C::NK];
  load_stationary<H, DH>(qf, q + ((long long)b * N + qrow) * C::D, g4);
  zero_pads<H, DH>(Kc, CK * 16, tid, WPB * 64);
  const int nchunks = (ntiles + CK - 1) / CK;
  float mx[H], sm[H];
#pragma unroll
  for (int h = 0; h < H; ++h) { mx[h] = -3.0e38f; sm[h] = 0.f; }
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Kc;
  st_Kc.fetch(kb, min(CK, ntiles) * 16, tid);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    __syncthreads();
    st_Kc.commit(Kc, nt * 16, tid);
    __syncthreads();
    {
      const int cn = ch + 1 < nchunks ? ch + 1 : ch;
      st_Kc.fetch(kb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
    }
    if (active)
      for (int kc = (KS == 1 ? 0 : ((ch * CK) & 1) ^ ksh); kc < nt; kc += KS) {
        f32x4 acc[H];
        tile_logits<H, DH>(acc, Kc, kc, qf, l15, g4);
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float tm = fmaxf(fmaxf(acc[h][0], acc[h][1]), fmaxf(acc[h][2], acc[h][3])) * c;      // c > 0
          const float mn = fmaxf(mx[h], tm);
          float a = sm[h] * fexp2(mx[h] - mn);
#pragma unroll
          for (int r = 0; r < 4; ++r) a += fexp2(fmaf(acc[h][r], c, -mn));
          sm[h] = a;
          mx[h] = mn;
        }
      }
  }
  if constexpr (KS == 2) {                  // merge the odd wave's running (max, sum) into the even wave's
    __syncthreads();
    float* cs = reinterpret_cast<float*>(smem_raw) + (wave >> 1) * (2 * H * 64);
    if (ksh) {
#pragma unroll
      for (int h = 0; h < H; ++h) { cs[h * 64 + lane] = mx[h]; cs[(H + h) * 64 + lane] = sm[h]; }
    }
    __syncthreads();
    if (!ksh) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float m2 = cs[h * 64 + lane], s2 = cs[(H + h) * 64 + lane];
        const float mn = fmaxf(mx[h], m2);
        sm[h] = sm[h] * fexp2(mx[h] - mn) + s2 * fexp2(m2 - mn);
        mx[h] = mn;
      }
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float m = mx[h], s_ = sm[h];
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s_, o, 64);
      const float mn = fmaxf(m, m2);
      s_ = s_ * fexp2(m - mn) + s2 * fexp2(m2 - mn);
      m = mn;
    }
    if (active && g4 == 0 && ksh == 0) {
      const float l = m + log2f(s_);
      lse2[((long long)b * H + h) * N + qrow] = l;
      rinv[((long long)b * H + h) * N + qrow] = 1.0f / (s_ * fexp2(m - l));
    }
  }
}

// column sum of a [nblocks][ncol] partial array over the rows rl, rl + step, ...: four independent accumulators over
// loads issued together (a single dependent chain pays one L2 round trip per row: 30 us for 832 rows)
__device__ __forceinline__ double strided_colsum(const float* __restrict__ p, int nblocks, int ncol, int col, int rl, int step) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int i = rl;
  for (; i + 3 * step < nblocks; i += 4 * step) {
    const float x0 = p[(long long)i * ncol + col], x1 = p[(long long)(i + step) * ncol + col];
    const float x2 = p[(long long)(i + 2 * step) * ncol + col], x3 = p[(long long)(i + 3 * step) * ncol + col];
    a0 += (double)x0; a1 += (double)x1; a2 += (double)x2; a3 += (double)x3;
  }
  for (; i < nblocks; i += step) a0 += (double)p[(long long)i * ncol + col];
  return (a0 + a1) + (a2 + a3);
}

// fp64 finalize: moments -> BatchNorm statistics of the mixed maps, running statistics, folded tables.
// stats layout: vu_kernels.h (VU_BN_STATS_*), extended by FWk = gamma rstd W / keep and XK = rstd W / keep.
// `direct`: the partial rows hold the moments of the mixed maps themselves (v2: [sum (A_g - shift_g)] [sum (A_g - shift_g)^2],
// shift_g = sum_h W[g,h] / N, the bias c_g not included) instead of the first / cross moments of the probabilities.
__global__ __launch_bounds__(1024) void flash_bn_finalize_kernel(const float* __restrict__ partials, int nblocks, const float* __restrict__ W,
                                         const float* __restrict__ cb, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* run_mean, float* run_var, float* stats, int H, int N, double count, int training,
                                         float momentum, float eps, float inv_keep, int direct) {
  __shared__ double mom[64];
  __shared__ double sd[1024];
  const int NM = direct ? 2 * H : H + H * (H + 1) / 2;
  const int tid = threadIdx.x;          // 1024 threads: COLS moment columns side by side (coalesced), 1024 / COLS row lanes
  if (training) {
    const int COLS = NM <= 16 ? 16 : 64, RL = 1024 / COLS;      // (the v2 forward has 16 columns: 64 row lanes, 13 rows each)
    const int col = tid & (COLS - 1), rl = tid / COLS;
    double a = 0.0;
    if (col < NM) a = strided_colsum(partials, nblocks, NM, col, rl, RL);
    sd[rl * COLS + col] = a;
    __syncthreads();
    if (tid < NM) {
      double t = 0.0;
      for (int r = 0; r < RL; ++r) t += sd[r * COLS + tid];
      mom[tid] = t / count;
    }
  }
  __syncthreads();
  if (tid < H) {
    const int g = tid;
    double mean, var;
    if (training && direct) {
      double sw = 0.0;
      for (int h = 0; h < H; ++h) sw += (double)W[g * H + h];
      const double d1 = mom[g], d2 = mom[H + g];
      mean = (double)cb[g] + sw / (double)N + d1;
      var = d2 - d1 * d1;
      if (var < 0.0) var = 0.0;
      const double unb = count > 1.0 ? count / (count - 1.0) : 1.0;
      run_mean[g] = (1.f - momentum) * run_mean[g] + momentum * (float)mean;
      run_var[g] = (1.f - momentum) * run_var[g] + momentum * (float)(var * unb);
    } else if (training) {
      const double cen = 1.0 / (double)N;
      double mu = (double)cb[g], vv = 0.0;
      for (int h = 0; h < H; ++h) mu += (double)W[g * H + h] * (mom[h] + cen);
      for (int h = 0; h < H; ++h)
        for (int h2 = 0; h2 < H; ++h2) {
          const int hi = h > h2 ? h : h2, lo = h > h2 ? h2 : h;
          const double cov = mom[H + hi * (hi + 1) / 2 + lo] - mom[h] * mom[h2];
          vv += (double)W[g * H + h] * (double)W[g * H + h2] * cov;
        }
      mean = mu; var = vv > 0.0 ? vv : 0.0;
      const double unb = count > 1.0 ? count / (count - 1.0) : 1.0;
      run_mean[g] = (1.f - momentum) * run_mean[g] + momentum * (float)mean;
      run_var[g] = (1.f - momentum) * run_var[g] + momentum * (float)(var * unb);
    } else {
      mean = run_mean[g]; var = run_var[g];
    }
    const float rstd = rsqrtf((float)var + eps);
    const float sc = gamma[g] * rstd;
    for (int h = 0; h < H; ++h) {
      stats[g * H + h] = W[g * H + h] * sc;
      stats[VU_BN_STATS_FWK(H) + g * H + h] = W[g * H + h] * sc * inv_keep;
      stats[VU_BN_STATS_XK(H) + g * H + h] = W[g * H + h] * rstd * inv_keep;
    }
    stats[H * H + g] = (cb[g] - (float)mean) * sc + beta[g];
    stats[H * H + H + g] = (float)mean;
    stats[H * H + 2 * H + g] = rstd;
    stats[H * H + 3 * H + g] = 0.f;
    stats[H * H + 4 * H + g] = 0.f;
    stats[VU_BN_STATS_SC(H) + g] = sc;
    stats[VU_BN_STATS_SC(H) + H + g] = (cb[g] - (float)mean) * sc + beta[g];
  }
}

// =============================================================================================
// apply pass: O = A^ v
// =============================================================================================
template <int H, int DH, int WPB, int CK>
__global__ __launch_bounds__(WPB * 64, 2) void flash_apply_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const float* __restrict__ lse2,
    const float* __restrict__ stats, bf16_t* __restrict__ O, int B, int N, float c, vu_rng rng_in) {
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);                       // [CK*16][PITCH]
  bf16_t* Vc = Kc + CK * 16 * C::PITCH;
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int ntiles = N >> 4, per = (ntiles + WPB - 1) / WPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * WPB + wave;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  const bf16_t* vb = v + (long long)b * N * C::D;
  bf16x8 qf[H * C::NK];
  load_stationary<H, DH>(qf, q + ((long long)b * N + qrow) * C::D, g4);
  float lse[H];
#pragma unroll
  for (int h = 0; h < H; ++h) lse[h] = lse2[((long long)b * H + h) * N + qrow];
  zero_pads<H, DH>(Kc, 2 * CK * 16, tid, WPB * 64);
  float* tab = reinterpret_cast<float*>(Vc + CK * 16 * C::PITCH);       // [H*H] gamma rstd W / keep, [H] folded bias
  for (int i = tid; i < H * H + H; i += WPB * 64) tab[i] = i < H * H ? stats[VU_BN_STATS_FWK(H) + i] : stats[H * H + (i - H * H)];
  f32x4 oacc[H][C::DT];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oacc[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);          // quad words per (sample, head) map
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  const int nchunks = (ntiles + CK - 1) / CK;

  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    if (!(VU_V1_NOSTAGE && ch > 0)) {
    __syncthreads();
    load_chunk<H, DH>(Kc, kb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    load_chunk<H, DH>(Vc, vb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    __syncthreads();
    }
    if (active)
      for (int kc = 0; kc < nt; ++kc) {
        f32x4 acc[H];
        tile_logits<H, DH>(acc, Kc, kc, qf, l15, g4);
        const uint32_t wt = wlane + 4u * (uint32_t)(ch * CK + kc);
        const uint32_t base = keep4(rng, wt).w, base_sh = base >> 12;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          keep4_t kp;
          kp.thr = rng.thr;
          kp.w = rng.thr ? vu_quad_head(base, base_sh, h) : 0u;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fexp2(fmaf(acc[h][r], c, -lse[h]));
            acc[h][r] = kept(kp, r) ? p : 0.f;
          }
        }
        LDS_FENCE();
        RowW<H> wcur = ld_row<H>(tab);
        float ccur = tab[H * H];
#pragma unroll
        for (int g = 0; g < H; ++g) {       // A^_g = folded bias + sum_h (gamma rstd W / keep)[g,h] P^_h, then O_g^T += V_g^T A^_g^T
          RowW<H> wnxt = wcur;
          float cnxt = ccur;
          if (g + 1 < H) { wnxt = ld_row<H>(tab + (g + 1) * H); cnxt = tab[H * H + g + 1]; }
          s16x4 aop[C::DT];
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) aop[dt] = tr_operand<C::PITCH>(Vc, kc * 16, g * DH + 16 * dt, l15, g4);
          LDS_FENCE();
          f32x4 a;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = ccur;
#pragma unroll
            for (int h = 0; h < H; ++h) x = fmaf(wcur.w[h], acc[h][r], x);
            a[r] = x;
          }
          const s16x4 bop = pack4s(a);
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) oacc[g][dt] = mfma16(aop[dt], bop, oacc[g][dt]);
          wcur = wnxt; ccur = cnxt;
        }
      }
  }
  if (active) {
    bf16_t* orow = O + ((long long)b * N + qrow) * C::D;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int f = 16 * dt + 4 * g4;                               // accumulator row = head feature
        if (f < DH) *reinterpret_cast<bf16x4*>(orow + h * DH + f) = pack4(oacc[h][dt]);
      }
  }
}

// =============================================================================================
// backward: recompute sweeps.  With P^ = mask * P (p~ = P^ / keep), x^_g = Xc_g + sum_h XK[g,h] P^_h,
//   e_g  = dA^_g - m1_g - m2_g x^_g                (dA_g = gamma_g rstd_g e_g; m1 = mean dA^, m2 = mean dA^ x^: bn_bwd_small)
//   dP_h = sum_g FWk[g,h] e_g                       (= dL/dP~_h / keep)
//   delta_h = sum_j P^_h dP_h ;  dS_h = P^_h dP_h - P_h delta_h ;  dq = scale dS k ; dk = scale dS^T q ; dv = A^^T dO
//   dW[g,h] = gamma_g rstd_g / keep * sum e_g P^_h ; dc_g = gamma_g rstd_g sum e_g
// dA^ = dO v^T is recomputed per tile as a second logits-shaped MFMA product (A = V rows, B = dO fragments) whose
// accumulator starts at -m1_g - m2_g Xc_g, so that e_g is reached by accumulating -m2_g XK[g,h] P^_h onto it head by head.
// The probabilities are kept sign-tagged in the logits registers (+p kept, -p dropped): P^ = max(tag, 0), P = |tag|.
// =============================================================================================
template <int H> struct BwdTab {
  float XK2T[H * H];   // [h][g] = -m2_g rstd_g W[g,h] / keep
  float FWkT[H * H];   // [h][g] = gamma_g rstd_g W[g,h] / keep
  float cin[H];        // -m1_g - m2_g Xc_g
};
template <int H>
__device__ __forceinline__ void load_bwd_tab(BwdTab<H>* tb, const float* __restrict__ stats, int tid, int nthr) {
  for (int i = tid; i < H * H; i += nthr) {
    const int h = i / H, g = i % H;
    const float m2 = stats[H * H + 4 * H + g];
    tb->XK2T[i] = -m2 * stats[VU_BN_STATS_XK(H) + g * H + h];
    tb->FWkT[i] = stats[VU_BN_STATS_FWK(H) + g * H + h];
  }
  for (int g = tid; g < H; g += nthr)
    tb->cin[g] = -stats[H * H + 3 * H + g] - stats[H * H + 4 * H + g] * stats[2 * H * H + 5 * H + g];
}

// logits-shaped MFMA product with a per-head constant as the initial accumulator; STREAM_A: the LDS chunk is the A
// operand (q-major sweeps: rows = keys) else the B operand (key-major sweeps: columns = queries).  The stationary
// operand (the wave's own 16 tokens) is either a register array `sf` or, to keep the register budget of the backward
// sweeps at two waves per SIMD, the wave's 16-row LDS image `sl` (k-slots >= DH / 8 zeroed after the read).
template <int H, int DH, bool STREAM_A>
__device__ __forceinline__ void tile_prod(f32x4 (&acc)[H], const bf16_t* Xc, int kc, const bf16x8* sf, const bf16_t* sl, const float* cin,
                                          int l15, int g4) {
  typedef FC<H, DH> C;
  if constexpr (C::NK == 1) {      // (verbatim single-product form, see tile_logits)
    const int gk = g4 < C::KS ? g4 : C::KS - 1;
    const bf16_t* xrow = Xc + (kc * 16 + l15) * C::PITCH + 8 * gk;
    // k-slots >= KS of the stationary operand must be zero: those lanes read the row's zeroed 16-byte pad for every head (a
    // per-lane head stride of 0) instead of a valid slot followed by 4 selects per head (32 VALU instructions per product)
    const bool dead = C::KS < 4 && g4 >= C::KS;
    const int hs = dead ? 0 : DH;
    const bf16_t* srow = sl ? sl + l15 * C::PITCH + (dead ? C::D : 8 * g4) : nullptr;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + h * DH);
      bf16x8 st;
      if (sl) {
        st = *reinterpret_cast<const bf16x8*>(srow + h * hs);
      } else st = sf[h];
      const float ci = cin ? cin[h] : 0.f;
      const f32x4 c0 = {ci, ci, ci, ci};
      acc[h] = STREAM_A ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, st, c0, 0, 0, 0)
                        : __builtin_amdgcn_mfma_f32_16x16x32_bf16(st, xf, c0, 0, 0, 0);
    }
    return;
  }
  const bf16_t* xrow0 = Xc + (kc * 16 + l15) * C::PITCH;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float ci = cin ? cin[h] : 0.f;
    f32x4 a = {ci, ci, ci, ci};
#pragma unroll
    for (int kh = 0; kh < C::NK; ++kh) {
      const int ks = C::KS - 4 * kh < 4 ? C::KS - 4 * kh : 4;   // valid 8-feature slots of this k-step
      const int gk = g4 < ks ? g4 : ks - 1;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow0 + h * DH + 32 * kh + 8 * gk);
      // k-slots >= ks of the stationary operand must be zero: those lanes read the row's zeroed 16-byte pad for every head (a
      // per-lane head stride of 0) instead of a valid slot followed by 4 selects per head (32 VALU instructions per product)
      const bool dead = ks < 4 && g4 >= ks;
      bf16x8 st;
      if (sl) st = *reinterpret_cast<const bf16x8*>(sl + l15 * C::PITCH + (dead ? C::D : h * DH + 32 * kh + 8 * g4));
      else st = sf[h * C::NK + kh];
      a = STREAM_A ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, st, a, 0, 0, 0)
                   : __builtin_amdgcn_mfma_f32_16x16x32_bf16(st, xf, a, 0, 0, 0);
    }
    acc[h] = a;
  }
}
// the wave's own 16 token rows -> its LDS image [16][PITCH] (wave-private: no barrier, LDS operations of a wave are ordered)
template <int H, int DH>
__device__ __forceinline__ void stage_own_rows(bf16_t* dst, const bf16_t* __restrict__ src, int lane) {
  typedef FC<H, DH> C;
  for (int v = lane; v < 16 * C::VPR; v += 64) {
    const int r = v / C::VPR, c = v - r * C::VPR;
    *reinterpret_cast<uint4*>(dst + r * C::PITCH + c * 8) = *reinterpret_cast<const uint4*>(src + (long long)r * C::D + c * 8);
  }
}

// logits -> sign-tagged probabilities, in place
// ZERO: dropped entries become 0 instead of -p, for sweeps that only need the kept probabilities P~ (no packed ReLU after)
template <int H, bool ZERO = false>
__device__ __forceinline__ void tag_probs(f32x4 (&S)[H], const float (&lse)[H], float c, const vu_rng& rng, uint32_t wt, uint32_t hstride) {
  const uint32_t base = keep4(rng, wt).w, base_sh = base >> 12;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    keep4_t kp;
    kp.thr = rng.thr;
    kp.w = rng.thr ? vu_quad_head(base, base_sh, h) : 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = fexp2(fmaf(S[h][r], c, -lse[h]));
      S[h][r] = kept(kp, r) ? p : (ZERO ? 0.f : -p);
    }
  }
}
// dA^' (accumulator started at cin) -> e, in place: e_g += sum_h XK2T[h][g] P^_h
template <int H>
__device__ __forceinline__ void mix_to_e(f32x4 (&E)[H], const f32x4 (&S)[H], const BwdTab<H>* tb) {
  LDS_FENCE();
  RowW<H> wcur = ld_row<H>(tb->XK2T);
#pragma unroll
  for (int h = 0; h < H; ++h) {
    RowW<H> wnxt = wcur;
    if (h + 1 < H) wnxt = ld_row<H>(tb->XK2T + (h + 1) * H);
    LDS_FENCE();
    f32x4 ph;
#pragma unroll
    for (int r = 0; r < 4; ++r) ph[r] = fmaxf(S[h][r], 0.f);
#pragma unroll
    for (int g = 0; g < H; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) E[g][r] = fmaf(wcur.w[g], ph[r], E[g][r]);
    wcur = wnxt;
  }
}
// dP_h = sum_g FWkT[h][g] e_g with the (prefetched) table row w
template <int H>
__device__ __forceinline__ f32x4 mix_back(const f32x4 (&E)[H], const RowW<H>& w) {
  f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < H; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) dp[r] = fmaf(w.w[g], E[g][r], dp[r]);
  return dp;
}

// ---- sweep 1 (q-major): delta_h[i] and the head-mix gradient sums ---------------------------------------------------
// DQ (round 4; the training backward when the forward saved V = sum_k bf16(P) k): dq leaves the SAME sweep.  dS = P~ dP~ - P delta
// needs delta of the whole row, so the sweep accumulates U = sum_k bf16(u) k (u = P~ dP~) and uses V from the forward:
// dq = scale (U - delta~ V) with delta~ = sum_k bf16(u) * rinv, rinv = 1 / sum_k bf16(P): the coefficients bf16(u) - delta~ bf16(P)
// sum to zero exactly, as the row-wise softmax backward requires (row_norm_note).  The delta written for the dk sweep stays
// sum_k u * rinv.  One whole chain rebuild (flash_bwd_dq_kernel) less per module.
template <int H, int DH, int WPB, int CK, bool DQ = false>
__global__ __launch_bounds__(WPB * 64, VU_FLASH_V1_WAVES(H, DH)) void flash_bwd_delta_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse2, const float* __restrict__ rinv, const float* __restrict__ stats, float* __restrict__ delta,
    float* __restrict__ partials, int B, int N, float c, vu_rng rng_in, const float* __restrict__ pkv = nullptr,
    bf16_t* __restrict__ dq = nullptr, float scale = 0.f, const float* __restrict__ rinv_b = nullptr, int lds_clear = 0) {
  typedef FC<H, DH> C;
  constexpr int NT = H * H + H;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (lds_clear) {            // diagnostic (VU_FLASH_V1_CLEAR=1): no value of the sweep may depend on what the LDS held before
    for (int i = threadIdx.x; i < lds_clear / 4; i += WPB * 64) reinterpret_cast<unsigned*>(smem_raw)[i] = 0x7f7f7f7fu;
    __syncthreads();
  }
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Vc = Kc + CK * 16 * C::PITCH;
  bf16_t* Qs = Vc + CK * 16 * C::PITCH + (threadIdx.x >> 6) * 32 * C::PITCH;             // this wave's q rows, then its dO rows
  bf16_t* dOs = Qs + 16 * C::PITCH;
  BwdTab<H>* tb = reinterpret_cast<BwdTab<H>*>(Vc + CK * 16 * C::PITCH + WPB * 32 * C::PITCH);
  float* red = reinterpret_cast<float*>(tb + 1);                          // [WPB][NT]
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int ntiles = N >> 4, per = (ntiles + WPB - 1) / WPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * WPB + wave;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  const bf16_t* vb = v + (long long)b * N * C::D;
  stage_own_rows<H, DH>(Qs, q + ((long long)b * N + tq * 16) * C::D, lane);
  stage_own_rows<H, DH>(dOs, dO + ((long long)b * N + tq * 16) * C::D, lane);
  float lse[H], dl[H], Tc[H], T[H * H];
#pragma unroll
  for (int h = 0; h < H; ++h) { lse[h] = lse2[((long long)b * H + h) * N + qrow]; dl[h] = 0.f; Tc[h] = 0.f; }
#pragma unroll
  for (int i = 0; i < H * H; ++i) T[i] = 0.f;
  float dlb[DQ ? H : 1], rsb[DQ ? H : 1];
  f32x4 dqa[DQ ? H : 1][C::DT];
  if constexpr (DQ) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      dlb[h] = 0.f;
      rsb[h] = rinv_b[((long long)b * H + h) * N + qrow];       // 1 / sum_k bf16(P) of the forward's moments sweep
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) dqa[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  zero_pads<H, DH>(Kc, 2 * CK * 16 + WPB * 32, tid, WPB * 64);
  load_bwd_tab<H>(tb, stats, tid, WPB * 64);
  VU_V1_FW_DECL;
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);          // quad words per (sample, head) map
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  const int nchunks = (ntiles + CK - 1) / CK;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    if (!(VU_V1_NOSTAGE && ch > 0)) {
    __syncthreads();
    load_chunk<H, DH>(Kc, kb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    load_chunk<H, DH>(Vc, vb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    __syncthreads();
    }
    if (active)
      for (int kc = 0; kc < nt; ++kc) {
        f32x4 S[H], E[H];
        tile_prod<H, DH, true>(S, Kc, kc, nullptr, Qs, nullptr, l15, g4);
        asm volatile("" ::: "memory");
        tile_prod<H, DH, true>(E, Vc, kc, nullptr, dOs, tb->cin, l15, g4);
        tag_probs<H>(S, lse, c, rng, wlane + 4u * (uint32_t)(ch * CK + kc), hstride);
        mix_to_e<H>(E, S, tb);
#pragma unroll
        for (int g = 0; g < H; ++g) Tc[g] += (E[g][0] + E[g][1]) + (E[g][2] + E[g][3]);
        LDS_FENCE();
        RowW<H> wcur = VU_V1_FW_ROW(0);
#pragma unroll
        for (int h = 0; h < H; ++h) {
          RowW<H> wnxt = wcur;
          if (h + 1 < H) wnxt = VU_V1_FW_ROW(h + 1);
          LDS_FENCE();
          s16x4 aop[C::DT];
          if constexpr (DQ) {
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) aop[dt] = tr_operand<C::PITCH>(Kc, kc * 16, h * DH + 16 * dt, l15, g4);
            LDS_FENCE();
          }
          const f32x4 dp = mix_back<H>(E, wcur);
          f32x4 u;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ph = fmaxf(S[h][r], 0.f);
            u[r] = ph * dp[r];
            dl[h] += u[r];
#pragma unroll
            for (int g = 0; g < H; ++g) T[g * H + h] = fmaf(E[g][r], ph, T[g * H + h]);
          }
          if constexpr (DQ) {
            const s16x4 bop = pack4s(u);
            const u32x2_t bw = __builtin_bit_cast(u32x2_t, bop);
            dlb[h] = sum2_bf16(bw[1], sum2_bf16(bw[0], dlb[h]));
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) dqa[h][dt] = mfma16(aop[dt], bop, dqa[h][dt]);       // U_h^T += K_h^T bf16(u_h)^T
          }
          wcur = wnxt;
        }
      }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    dl[h] += __shfl_xor(dl[h], 16, 64);
    dl[h] += __shfl_xor(dl[h], 32, 64);
    const float rs = rinv[((long long)b * H + h) * N + qrow];                                                               // row_norm_note
    if (active && g4 == 0) delta[((long long)b * H + h) * N + qrow] = dl[h] * rs;
    if constexpr (DQ) {
      float db = dlb[h];
      db += __shfl_xor(db, 16, 64);
      db += __shfl_xor(db, 32, 64);
      const float dt_ = db * rsb[h];                                               // delta~ of (query l15, head h): in every lane group
      if (active) {
        bf16_t* orow = dq + ((long long)b * N + qrow) * C::D;
        const float* vrow = pkv + ((long long)b * N + qrow) * C::D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          const int f = 16 * dt + 4 * g4;                                           // accumulator row = head feature, column = query
          if (f < DH) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(vrow + h * DH + f);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = scale * fmaf(-dt_, pv[r], dqa[h][dt][r]);
            *reinterpret_cast<bf16x4*>(orow + h * DH + f) = pack4(o);
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < H * H; ++i) { const float x = vu_wave_sum(active ? T[i] : 0.f); if (lane == 0) red[wave * NT + i] = x; }
#pragma unroll
  for (int g = 0; g < H; ++g) { const float x = vu_wave_sum(active ? Tc[g] : 0.f); if (lane == 0) red[wave * NT + H * H + g] = x; }
  __syncthreads();
  for (int i = tid; i < NT; i += WPB * 64) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) a += red[w * NT + i];
    partials[(long long)blockIdx.x * NT + i] = a;
  }
}

// head-mix gradients from the sweep-1 partial sums
__global__ __launch_bounds__(256) void flash_bwd_mix_finalize_kernel(const float* __restrict__ partials, int nblocks,
                                                                      const float* __restrict__ stats, float* dW, float* dc, int H,
                                                                      float inv_keep) {
  // one wave per column of the (nblocks x NT) partials, 4 columns per workgroup; fixed order: lane l adds rows l, l + 64, ...
  // in fp64, then a shuffle tree.  (One workgroup walking all 72 columns took 11.5 us.)  Round 6: 256-thread workgroups instead of
  // five of 1024 - this launch runs beside the low-priority dv sweep, whose resident workgroups leave no CU with 16 free wave slots:
  // the 1024-thread form sat 20 us in the dispatcher (28.9 us per launch in the step's trace for a 240 KB reduction) and the dk
  // sweep behind it waited; the sums and their order are unchanged.
  const int NT = H * H + H;
  const int c2 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c2 >= NT) return;          // wave-uniform
  // (eight independent accumulators over loads issued together: as ONE dependent chain the 26 strided rows of a lane at 64 images
  // were 26 memory round trips - 22.9 us per launch)
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int r = lane;
  for (; r + 7 * 64 < nblocks; r += 8 * 64) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = partials[(long long)(r + 64 * u) * NT + c2];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += (double)x[u];
  }
  for (; r < nblocks; r += 64) a[0] += (double)partials[(long long)r * NT + c2];
  double t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
  if (lane == 0) {
    const int g = c2 < H * H ? c2 / H : c2 - H * H;
    const float gs = stats[2 * H * H + 6 * H + g];               // gamma rstd
    if (c2 < H * H) dW[c2] += (float)(t * gs * inv_keep); else dc[g] += (float)(t * gs);
  }
}

// ---- sweep 2 (q-major): dq ------------------------------------------------------------------------------------------
template <int H, int DH, int WPB, int CK>
__global__ __launch_bounds__(WPB * 64, 2) void flash_bwd_dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse2, const float* __restrict__ delta, const float* __restrict__ stats, bf16_t* __restrict__ dq,
    int B, int N, float c, float scale, vu_rng rng_in) {
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Vc = Kc + CK * 16 * C::PITCH;
  bf16_t* Qs = Vc + CK * 16 * C::PITCH + (threadIdx.x >> 6) * 32 * C::PITCH;             // this wave's q rows, then its dO rows
  bf16_t* dOs = Qs + 16 * C::PITCH;
  BwdTab<H>* tb = reinterpret_cast<BwdTab<H>*>(Vc + CK * 16 * C::PITCH + WPB * 32 * C::PITCH);
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int ntiles = N >> 4, per = (ntiles + WPB - 1) / WPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * WPB + wave;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  const bf16_t* vb = v + (long long)b * N * C::D;
  stage_own_rows<H, DH>(Qs, q + ((long long)b * N + tq * 16) * C::D, lane);
  stage_own_rows<H, DH>(dOs, dO + ((long long)b * N + tq * 16) * C::D, lane);
  // row constants of the wave's 16 queries: [2][H][16] floats, wave-private, re-read per tile (16 registers saved)
  float* rowc = reinterpret_cast<float*>(tb + 1) + wave * (2 * H * 16);
  for (int i = lane; i < 2 * H * 16; i += 64) {
    const int which = i / (H * 16), h = (i / 16) % H, j = i & 15;
    rowc[i] = (which ? delta : lse2)[((long long)b * H + h) * N + tq * 16 + j];
  }
  zero_pads<H, DH>(Kc, 2 * CK * 16 + WPB * 32, tid, WPB * 64);
  load_bwd_tab<H>(tb, stats, tid, WPB * 64);
  VU_V1_FW_DECL;
  f32x4 dqa[H][C::DT];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) dqa[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);          // quad words per (sample, head) map
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  const int nchunks = (ntiles + CK - 1) / CK;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    if (!(VU_V1_NOSTAGE && ch > 0)) {
    __syncthreads();
    load_chunk<H, DH>(Kc, kb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    load_chunk<H, DH>(Vc, vb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    __syncthreads();
    }
    if (active)
      for (int kc = 0; kc < nt; ++kc) {
        f32x4 S[H], E[H];
        tile_prod<H, DH, true>(S, Kc, kc, nullptr, Qs, nullptr, l15, g4);
        asm volatile("" ::: "memory");
        tile_prod<H, DH, true>(E, Vc, kc, nullptr, dOs, tb->cin, l15, g4);
        {
          float lse[H];
#pragma unroll
          for (int h = 0; h < H; ++h) lse[h] = rowc[h * 16 + l15];
          tag_probs<H>(S, lse, c, rng, wlane + 4u * (uint32_t)(ch * CK + kc), hstride);
        }
        mix_to_e<H>(E, S, tb);
        LDS_FENCE();
        RowW<H> wcur = VU_V1_FW_ROW(0);
#pragma unroll
        for (int h = 0; h < H; ++h) {
          RowW<H> wnxt = wcur;
          if (h + 1 < H) wnxt = VU_V1_FW_ROW(h + 1);
          const float dlh = rowc[(H + h) * 16 + l15];
          s16x4 aop[C::DT];
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) aop[dt] = tr_operand<C::PITCH>(Kc, kc * 16, h * DH + 16 * dt, l15, g4);
          LDS_FENCE();
          const f32x4 dp = mix_back<H>(E, wcur);
          f32x4 ds;
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[r] = fmaf(fmaxf(S[h][r], 0.f), dp[r], -fabsf(S[h][r]) * dlh);
          const s16x4 bop = pack4s(ds);
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) dqa[h][dt] = mfma16(aop[dt], bop, dqa[h][dt]);       // dq_h^T += K_h^T dS_h^T
          wcur = wnxt;
        }
      }
  }
  if (active) {
    bf16_t* orow = dq + ((long long)b * N + qrow) * C::D;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int f = 16 * dt + 4 * g4;
        if (f < DH) {
          f32x4 o = dqa[h][dt];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] *= scale;
          *reinterpret_cast<bf16x4*>(orow + h * DH + f) = pack4(o);
        }
      }
  }
}

// ---- sweeps 3 / 4 (key-major loop, same tile orientation): dk, dv ---------------------------------------------------
// A wave owns 16 keys (fragments in registers as the A operand) and walks every query tile (Q / dO chunks in LDS as
// the B operand), so the accumulator still has the query on the lane and 4 consecutive keys in its registers, and the
// tile body is the one of the q-major sweeps.  dk^T = Q^T dS and dv^T = dO^T A^ contract over the QUERY (lane) index:
// the packed tile goes through a 512-byte wave-private LDS image [query][key] and comes back through the transposing
// read as the B operand (k = query, column = key); the A operand is the transposing read of the Q / dO chunk.
template <int H, int DH, int WPB, int CK, bool DV>
__global__ __launch_bounds__(WPB * 64, VU_FLASH_V1_WAVES(H, DH)) void flash_bwd_dkv_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse2, const float* __restrict__ delta, const float* __restrict__ stats, bf16_t* __restrict__ out,
    int B, int N, float c, float scale, vu_rng rng_in, int lds_clear = 0) {
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (lds_clear) {            // diagnostic (VU_FLASH_V1_CLEAR=1), see flash_bwd_delta_kernel
    for (int i = threadIdx.x; i < lds_clear / 4; i += WPB * 64) reinterpret_cast<unsigned*>(smem_raw)[i] = 0x7f7f7f7fu;
    __syncthreads();
  }
  bf16_t* Qc = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Dc = Qc + CK * 16 * C::PITCH;                                   // dO chunk
  bf16_t* Ks = Dc + CK * 16 * C::PITCH + (threadIdx.x >> 6) * 32 * C::PITCH;             // dk form: this wave's k rows, then its v rows
  bf16_t* Vs = Ks + 16 * C::PITCH;
  float* lsec = reinterpret_cast<float*>(Dc + CK * 16 * C::PITCH + (DV ? 0 : WPB * 32 * C::PITCH));       // [H][CK*16]
  float* dlc = lsec + H * CK * 16;                                        // [H][CK*16]
  BwdTab<H>* tb = reinterpret_cast<BwdTab<H>*>(dlc + H * CK * 16);
  float* ftab = reinterpret_cast<float*>(tb + 1);                        // DV: FWk [g][h] (H*H), cf[H]
  bf16_t* img = reinterpret_cast<bf16_t*>(ftab + H * H + H) + (threadIdx.x >> 6) * 256;     // this wave's [16 q][16 keys]
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int ntiles = N >> 4, per = (ntiles + WPB - 1) / WPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * WPB + wave;
  const bool active = t < ntiles;
  const int tk = active ? t : ntiles - 1;
  const int krow = tk * 16 + l15;
  const bf16_t* qb = q + (long long)b * N * C::D;
  const bf16_t* dob = dO + (long long)b * N * C::D;
  bf16x8 kf[H * C::NK];
  if (DV) load_stationary<H, DH>(kf, k + ((long long)b * N + krow) * C::D, g4);
  else {
    stage_own_rows<H, DH>(Ks, k + ((long long)b * N + tk * 16) * C::D, lane);
    stage_own_rows<H, DH>(Vs, v + ((long long)b * N + tk * 16) * C::D, lane);
  }
  zero_pads<H, DH>(Qc, 2 * CK * 16 + (DV ? 0 : WPB * 32), tid, WPB * 64);
  load_bwd_tab<H>(tb, stats, tid, WPB * 64);
  VU_V1_FW_DECL;
  for (int i = tid; i < H * H + H; i += WPB * 64) ftab[i] = i < H * H ? stats[VU_BN_STATS_FWK(H) + i] : stats[H * H + (i - H * H)];
  f32x4 oa[H][C::DT];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oa[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);
  // quad-word index of map element (row = (b H + h) N + query, key 16 tk + 4 g4): the query part is added per tile
  const uint32_t wkey = (uint32_t)(((unsigned long long)b * N * (unsigned long long)N) >> 2) + 4u * (uint32_t)tk + (uint32_t)g4;
  const uint32_t wq = (uint32_t)(N >> 2);                                  // quad words per map row
  const int nchunks = (ntiles + CK - 1) / CK;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    __syncthreads();
    load_chunk<H, DH>(Qc, qb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    load_chunk<H, DH>(Dc, dob + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    for (int i = tid; i < H * nt * 16; i += WPB * 64) {
      const int h = i / (nt * 16), j = i - h * (nt * 16);
      lsec[h * CK * 16 + j] = lse2[((long long)b * H + h) * N + ch * CK * 16 + j];
      dlc[h * CK * 16 + j] = delta ? delta[((long long)b * H + h) * N + ch * CK * 16 + j] : 0.f;
    }
    __syncthreads();
    if (active)
      for (int qc = 0; qc < nt; ++qc) {
        float lse[H], dl[H];
#pragma unroll
        for (int h = 0; h < H; ++h) { lse[h] = lsec[h * CK * 16 + qc * 16 + l15]; dl[h] = dlc[h * CK * 16 + qc * 16 + l15]; }
        f32x4 S[H];
        tile_prod<H, DH, false>(S, Qc, qc, DV ? kf : nullptr, DV ? nullptr : Ks, nullptr, l15, g4);
        const uint32_t wt = wkey + (uint32_t)((ch * CK + qc) * 16 + l15) * wq;
        tag_probs<H>(S, lse, c, rng, wt, hstride);
        if constexpr (DV) {
          LDS_FENCE();
          RowW<H> wcur = ld_row<H>(ftab);
          float ccur = ftab[H * H];
#pragma unroll
          for (int g = 0; g < H; ++g) {       // A^_g, then dv_g^T += dO_g^T A^_g
            RowW<H> wnxt = wcur;
            float cnxt = ccur;
            if (g + 1 < H) { wnxt = ld_row<H>(ftab + (g + 1) * H); cnxt = ftab[H * H + g + 1]; }
            s16x4 aop[C::DT];
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) aop[dt] = tr_operand<C::PITCH>(Dc, qc * 16, g * DH + 16 * dt, l15, g4);
            LDS_FENCE();
            f32x4 a;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x = ccur;
#pragma unroll
              for (int h = 0; h < H; ++h) x = fmaf(wcur.w[h], fmaxf(S[h][r], 0.f), x);
              a[r] = x;
            }
            *reinterpret_cast<s16x4*>(img + l15 * 16 + 4 * g4) = pack4s(a);
            const s16x4 bop = tr_operand<16>(img, 0, 0, l15, g4);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) oa[g][dt] = mfma16(aop[dt], bop, oa[g][dt]);
            wcur = wnxt; ccur = cnxt;
          }
        } else {
          f32x4 E[H];
          asm volatile("" ::: "memory");
          tile_prod<H, DH, false>(E, Dc, qc, nullptr, Vs, tb->cin, l15, g4);
          mix_to_e<H>(E, S, tb);
          LDS_FENCE();
          RowW<H> wcur = VU_V1_FW_ROW(0);
#pragma unroll
          for (int h = 0; h < H; ++h) {
            RowW<H> wnxt = wcur;
            if (h + 1 < H) wnxt = VU_V1_FW_ROW(h + 1);
            s16x4 aop[C::DT];
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) aop[dt] = tr_operand<C::PITCH>(Qc, qc * 16, h * DH + 16 * dt, l15, g4);
            LDS_FENCE();
            const f32x4 dp = mix_back<H>(E, wcur);
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[r] = fmaf(fmaxf(S[h][r], 0.f), dp[r], -fabsf(S[h][r]) * dl[h]);
            *reinterpret_cast<s16x4*>(img + l15 * 16 + 4 * g4) = pack4s(ds);
            const s16x4 bop = tr_operand<16>(img, 0, 0, l15, g4);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) oa[h][dt] = mfma16(aop[dt], bop, oa[h][dt]);       // dk_h^T += Q_h^T dS_h
            wcur = wnxt;
          }
        }
      }
  }
  if (active) {
    bf16_t* orow = out + ((long long)b * N + krow) * C::D;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int f = 16 * dt + 4 * g4;
        if (f < DH) {
          f32x4 o = oa[h][dt];
          if (!DV) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= scale;
          }
          *reinterpret_cast<bf16x4*>(orow + h * DH + f) = pack4(o);
        }
      }
  }
}

// =============================================================================================
// v2 tile body (8 heads): the two 8 x 8 head contractions run on the matrix cores.
//
// The first version above is bound by VALU issue (rocprofv3: 450 - 770 vector instructions per 16 x 16 x 8 tile, each
// ~4 cycles of a wave; profiles/r02a_flash_v1_*): of ~24 instructions per map element of the dq sweep, 8 are the
// fp32 head mixes.  Here the mixes are MFMAs on bf16-packed tiles, with the tables split hi + lo (two MFMAs) so that
// the coefficients keep 16 significant bits:
//   OL ("original layout", accumulator of K Q^T):  lane (q, g4): S[h][r] = head h, key 4 g4 + r
//   for each r: B operand = the lane's 8 heads packed (k = 8 g4 + h); A = block-structured table
//       A_half[row = (a', g)][k = (kg, h)] = M[g][h] * [kg == 2 half + a']
//   -> ML ("mixed layout"): lane (q, g4 = 2 a + hh): X[half][r][j] = head 4 hh + j, key 8 half + 4 a + r
// A lane of ML holds 4 heads at 8 keys instead of 8 heads at 4 keys.  Everything downstream of the first mix lives in
// ML: the identity table converts any OL tile (exactly, it is already bf16), the transposed mix is a 16x16x16 MFMA that
// maps ML to ML (k = (a, hh, j), row = (a', h)), and the products that contract over keys take, per (j, half), the
// lane's 4 keys as the B operand of a 16x16x16 MFMA whose A operand holds 8 features of head j (rows 0..7, from the
// lanes with hh = 0) and 8 features of head 4 + j (rows 8..15, hh = 1), the other half zero: its accumulator is 8
// features x 2 heads x 16 queries.  The zero half comes from a zeroed LDS region (an address select per tile, no
// per-operand masking).
// =============================================================================================
struct MixOp { bf16x8 hi[2], lo[2]; };       // [half]
struct BackOp { s16x4 hi, lo; };

__device__ __forceinline__ void split_hilo(float w, bf16_t& hi, bf16_t& lo) {
  hi = (bf16_t)w;
  lo = (bf16_t)(w - (float)hi);
}
// M: 8 x 8 row-major fp32 (M[g][h]) anywhere readable; scale[g] optional row factor
__device__ __forceinline__ void make_mix_op(MixOp& op, const float* M, const float* rowscale, int l15, int g4, float scale = 1.f) {
  const int ar = l15 >> 3, g = l15 & 7;
  const float rs = (rowscale ? rowscale[g] : 1.f) * scale;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const bool on = g4 == 2 * half + ar;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      bf16_t hi, lo;
      split_hilo(on ? M[g * 8 + h] * rs : 0.f, hi, lo);
      op.hi[half][h] = hi; op.lo[half][h] = lo;
    }
  }
}
__device__ __forceinline__ void make_identity_op(bf16x8 (&id)[2], int l15, int g4) {
  const int ar = l15 >> 3, g = l15 & 7;
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int h = 0; h < 8; ++h) id[half][h] = (bf16_t)((g4 == 2 * half + ar && h == g) ? 1.f : 0.f);
}
// transposed mix: A[row = (a', h)][k = 4 g4 + jj] = M[g = 4 (g4 & 1) + jj][h] * [(g4 >> 1) == a']
__device__ __forceinline__ void make_back_op(BackOp& op, const float* M, int l15, int g4) {
  const int ar = l15 >> 3, h = l15 & 7;
  bf16x4 hi, lo;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    bf16_t a, b;
    split_hilo(((g4 >> 1) == ar) ? M[(4 * (g4 & 1) + jj) * 8 + h] : 0.f, a, b);
    hi[jj] = a; lo[jj] = b;
  }
  op.hi = __builtin_bit_cast(s16x4, hi); op.lo = __builtin_bit_cast(s16x4, lo);
}

// the 8 heads of register r, packed: B operand of a mix MFMA
__device__ __forceinline__ void pack_heads(const f32x4 (&S)[8], bf16x8 (&pk)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const u32x4_t t = {pk2(S[0][r], S[1][r]), pk2(S[2][r], S[3][r]), pk2(S[4][r], S[5][r]), pk2(S[6][r], S[7][r])};
    pk[r] = __builtin_bit_cast(bf16x8, t);
  }
}
// negative (= dropped, sign-tagged) entries -> 0: bf16 bit patterns compared as signed 16-bit integers
__device__ __forceinline__ bf16x8 relu_packed(const bf16x8& x) {
  const s16x8 v = __builtin_bit_cast(s16x8, x);
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(v, z));
}
__device__ __forceinline__ f32x4 mfma32(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// Two K = 16 contractions into the same accumulator as ONE K = 32 instruction: the lane's four k-slots of each become its eight
// (any assignment of k-slots is fine as long as both operands use the same one).  v_mfma_f32_16x16x16_bf16 occupies the matrix
// pipe for the same 16 cycles as v_mfma_f32_16x16x32_bf16 (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = 16.0 on sweeps that mix both).
__device__ __forceinline__ bf16x8 join_k(const s16x4& a, const s16x4& b) {
  const s16x8 t = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, t);
}
// ML tile X[half][r] (f32x4 over j) = cin + (op.hi + op.lo) . pk[r]
__device__ __forceinline__ void mix_ml(f32x4 (&X)[2][4], const MixOp& op, const bf16x8 (&pk)[4], const f32x4& cin) {
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int r = 0; r < 4; ++r) X[half][r] = mfma32(op.lo[half], pk[r], mfma32(op.hi[half], pk[r], cin));
}
__device__ __forceinline__ void convert_ml(f32x4 (&X)[2][4], const bf16x8 (&id)[2], const bf16x8 (&pk)[4], bool accumulate) {
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int r = 0; r < 4; ++r) X[half][r] = mfma32(id[half], pk[r], accumulate ? X[half][r] : f32x4{0.f, 0.f, 0.f, 0.f});
}

// Per-lane source of the A operand of a key-contracting product (rows 0..7: 8 features of head j, rows 8..15: 8
// features of head 4 + j; k-slot 4 g4 + r = key 8 half + 4 (g4 >> 1) + r, live only where (g4 & 1) matches the row's
// head half).  Returns the lane's element offset into the chunk for tile kc (or into the zero region); the caller adds
// the compile-time part (8 half PITCH + j DH + fb 8) as an immediate.
template <int H, int DH>
struct TrSrc {
  int lane_off;      // live lanes: (4 (g4 >> 1) + qq) PITCH + 4 (pp >> 1) DH + 4 (pp & 1)
  bool live;
  int zero_off;      // dead lanes: start of the zero region (covers the largest immediate), relative to the chunk
  __device__ __forceinline__ void init(int l15, int g4, int zero_off_) {
    typedef FC<H, DH> C;
    const int qq = l15 >> 2, pp = l15 & 3;
    live = (pp >> 1) == (g4 & 1);
    lane_off = (4 * (g4 >> 1) + qq) * C::PITCH + 4 * (pp >> 1) * DH + 4 * (pp & 1);
    zero_off = zero_off_;
  }
  __device__ __forceinline__ int base(int kc) const { typedef FC<H, DH> C; return live ? lane_off + kc * 16 * C::PITCH : zero_off; }
};
template <int H, int DH>
__device__ __forceinline__ s16x4 tr_half(const bf16_t* Xc, int base, int half, int j, int fb) {
  typedef FC<H, DH> C;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Xc + base + 8 * half * C::PITCH + j * DH + 8 * fb));
}
template <int H, int DH> constexpr int tr_zero_elems() { return 8 * FC<H, DH>::PITCH + 3 * DH + 8 * ((DH + 7) / 8) + 8; }
// The same source with one base per key half: the immediates left (j DH + fb 8) stay inside a zero strip of
// tr_strip_elems() elements instead of a region that spans 8 rows (LDS is the scarce resource of the fused dq sweep).
template <int H, int DH>
struct TrSrc2 {
  int lane_off, zero_off;
  bool live;
  __device__ __forceinline__ void init(int l15, int g4, int zero_off_) {
    typedef FC<H, DH> C;
    const int qq = l15 >> 2, pp = l15 & 3;
    live = (pp >> 1) == (g4 & 1);
    lane_off = (4 * (g4 >> 1) + qq) * C::PITCH + 4 * (pp >> 1) * DH + 4 * (pp & 1);
    zero_off = zero_off_;
  }
  __device__ __forceinline__ int base(int kc, int half) const {
    typedef FC<H, DH> C;
    return live ? lane_off + (kc * 16 + 8 * half) * C::PITCH : zero_off;
  }
};
template <int DH>
__device__ __forceinline__ s16x4 tr_rel(const bf16_t* Xc, int base, int j, int fb) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Xc + base + j * DH + 8 * fb));
}
template <int DH> constexpr int tr_strip_elems() { return ((3 * DH + 8 * ((DH + 7) / 8) + 8 + 7) / 8) * 8; }


// ---- probability cache (round 5) --------------------------------------------------------------------------------------
// The training forward's moments sweep packs the sign-tagged probabilities of every 16 x 16 tile of ALL heads to bf16 anyway
// (pack_heads: the operand of the head mix).  With a cache (vu_flash_args::pcache) it also STORES those four registers per lane:
// tile (b, query tile qt, key tile kt) = 4 KB as [r][lane] 16-byte vectors - lane (query l15, key quad g4) holds the 8 heads of key
// 4 g4 + r - and the four sweeps after it (apply, dq + delta, dk, dv) stream the tile back instead of rebuilding logits -> exp2 ->
// mask hash -> compare / select -> pack (measured round 3: ~half of a sweep).  Same bits as the recomputed operand, so every
// result is identical to the recompute form's.  The query-major sweeps walk a query tile's row of tiles (contiguous), the
// key-major ones a column (4 KB pieces, stride N / 16 tiles).  2 bytes per map element per module: B h N^2 2 (Base level 2 at 64
// images: 630 MB), written once, read four times, non-temporal on both sides.
struct PTile { u32x4_t r[4]; };
// The two sweeps that do little besides streaming the cache (apply, dv) take their tiles through a wave-private LDS RING filled by
// LDS-DMA (global_load_lds_dwordx4, as in vu_bgemm.hip): a tile's [r][lane] layout is four linear 1 KiB pieces, exactly what one DMA
// instruction writes, so VU_PC_RING tiles are in flight per wave with no staging registers, and the waits are explicit
// (s_waitcnt vmcnt(4 * tiles younger than the one wanted)).  hipcc does not count these loads: wherever it waits for loads of its own
// (the register-staged chunk at a chunk boundary) it waits for the ring as well - correct, and the reason for longer chunks here.
// With register-staged tiles instead (round 5, first form) hipcc's own waits drained the prefetch at every loop head
// (s_waitcnt vmcnt(0) behind the reload): 144 us per sweep at 64 images where the bare stream takes 94 (tools/probe/tile_stream_probe).
#ifndef VU_PC_RING
#define VU_PC_RING 2
#endif
// one tile: four pieces of 1 KiB, global byte offsets v[r] = lane * 16 + r * 1024 from `tile`, LDS destinations 1 KiB apart (M0 carries
// the LDS address and is compiler-reserved: saved and restored inside the statement - the idiom of vu_bgemm.hip's dma16x4)
__device__ __forceinline__ void pc_dma4(const void* tile, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 nt\n\t"
               "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\t"
               "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1 nt\n\t"
               "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(tile), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_dst) : "memory", "scc");
}
// Instruction-group scheduling hint at the head of a tile body (__builtin_amdgcn_iglp_opt(0): hipcc interleaves the LDS reads with the
// matrix instructions of the region instead of its default order - a burst of ~260 vector instructions, then the MFMAs behind
// ~60 waits).  Measured at 64 images (round 5, two runs each on one box): cached dq sweep 339 - 342 -> 319 - 320 us (iglp_opt(1): 329);
// cached dk sweep 302 -> 299; moments 201 -> 205, apply / dv unchanged; -amdgpu-sched-strategy=max-ilp / max-memory-clause for the
// whole file: slower.  The other instantiations of the dq sweep (d = 8 / 32, the split forms) and the dk sweep at those shapes measured
// 3 - 10 % SLOWER with the hint, so it is on for the unsplit d = 24 dq sweep only (-DVU_IGLP_DQ=-1 turns it off).
#ifndef VU_IGLP_DQ
#define VU_IGLP_DQ 0
#endif
#ifndef VU_IGLP_DK
#define VU_IGLP_DK -1
#endif
#ifndef VU_IGLP_FW
#define VU_IGLP_FW -1
#endif
#define VU_IGLP_HINT(which) do { if constexpr ((which) >= 0) __builtin_amdgcn_iglp_opt((which) >= 0 ? (which) : 0); } while (0)
// (Explicit sched_group_barrier pipelines over the tile body - MFMA : VALU 1 : 3, MFMA : VALU : DS-read 1 : 2 : 1, 2 DS : 2 MFMA : 6 VALU :
// 1 DS-write - measured 5 - 7 % SLOWER than the hint on the dq sweep and 3 - 6 % slower than no hint on the dk sweep.)
#ifndef VU_DQX_ABLATE
#define VU_DQX_ABLATE 0         // measurement builds of the cached dq sweep: 1 no head-mix-gradient images, 2 no dq product, 4 no transposed mix, 8 no chunk staging / barriers after the first chunk, 16 no dA^ product / mix
#endif
#ifndef VU_PC_ABLATE
#define VU_PC_ABLATE 0          // measurement builds (results wrong by construction): 1 no cache stream, 2 no chunk staging / barriers after the first chunk, 4 no PV products
#endif
template <int RD>
struct PRing {
  const unsigned char* base;   // the wave's slot 0 (generic LDS pointer for the reads)
  unsigned lds0, v0;           // its LDS byte address; lane * 16
  int seq;                     // tiles taken so far
  __device__ __forceinline__ void init(unsigned char* ring, int wave, int lane) {
    base = ring + wave * (RD * 4096);
    lds0 = (unsigned)(size_t)base; v0 = (unsigned)lane * 16u; seq = 0;
  }
  __device__ __forceinline__ void issue(const void* pc, long long tile, int slot) const {
    if (VU_PC_ABLATE & 1) return;
    // (wave-uniform values that hipcc cannot prove uniform - they derive from threadIdx.x >> 6: made scalar explicitly)
    const unsigned long long a = (unsigned long long)(reinterpret_cast<const char*>(pc) + tile * 4096);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const void* tp = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
    pc_dma4(tp, v0, v0 + 1024u, v0 + 2048u, v0 + 3072u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot * 4096u));
  }
  // the oldest tile in flight -> registers (its RD - 1 successors stay in flight), then the slot is free for `refill`
  __device__ __forceinline__ int take(bf16x8 (&pk)[4], int lane) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RD - 1) * 4) : "memory");
    const int slot = seq % RD;
    const unsigned char* p = base + slot * 4096 + lane * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = *reinterpret_cast<const bf16x8*>(p + r * 1024);
    ++seq;
    return slot;
  }
  // after the registers of `take` have arrived (s_waitcnt lgkmcnt(0): a DMA write must not overtake the LDS reads of its slot)
  __device__ __forceinline__ void refill(const void* pc, long long tile, int slot) const {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue(pc, tile, slot);
  }
  __device__ __forceinline__ static void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};
__device__ __forceinline__ const u32x4_t* pc_tile(const void* pc, int b, int nt, int qt, int kt, int lane) {
  return reinterpret_cast<const u32x4_t*>(pc) + ((((long long)b * nt + qt) * nt + kt) * 4) * 64 + lane;
}
__device__ __forceinline__ void pc_load(PTile& t, const u32x4_t* p) {
#pragma unroll
  for (int r = 0; r < 4; ++r) t.r[r] = __builtin_nontemporal_load(p + r * 64);
}
__device__ __forceinline__ void pc_store(const bf16x8 (&pk)[4], u32x4_t* p) {
#pragma unroll
  for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, pk[r]), p + r * 64);
}
__device__ __forceinline__ void pc_take(bf16x8 (&pk)[4], const PTile& t) {
#pragma unroll
  for (int r = 0; r < 4; ++r) pk[r] = __builtin_bit_cast(bf16x8, t.r[r]);
}

// ---- stats pass, sweep 3 only (sweeps 1 / 2 are the v1 code): moments of the MIXED map directly -------------------
// per lane 4 heads x (sum, sum of squares) of A_g - shift_g, shift_g = sum_h W[g,h] / N (the exact mean without dropout)
template <int DH, int WPB, int CK, int KS = 1, bool PC = false>
__global__ __launch_bounds__(WPB * 64, 2) void flash2_moments_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                      const float* __restrict__ lse2, const float* __restrict__ W,
                                                                      float* __restrict__ partials, float* __restrict__ pk_out,
                                                                      float* __restrict__ rinv, int B, int N, float c, vu_rng rng_in,
                                                                      void* __restrict__ pcache) {
  constexpr int H = 8;
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);
  float* red = reinterpret_cast<float*>(Kc + CK * 16 * C::PITCH);        // [WPB][2 H]
  // s_b = sum_k bf16(P) (row_norm_note) for free where the head dim leaves spare rows in its last 16-feature block (d = 24:
  // rows 24..31; d = 8: rows 8..15): the lanes that supply those rows' addresses to the transposing read point at a strip
  // of ones instead of the next head's features, so the P k product accumulates sum_k bf16(P) * 1 in rows that were unused
  constexpr bool SPARE = 16 * C::DT > DH;
  constexpr int ONES = 7 * DH + 16;                                       // covers the per-head immediates of a spare lane
  bf16_t* ones = reinterpret_cast<bf16_t*>(red + WPB * 16);
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  if (SPARE) for (int i = tid; i < ONES; i += WPB * 64) ones[i] = (bf16_t)1.0f;
  constexpr int TPB = WPB / KS;
  const int ntiles = N >> 4, per = (ntiles + TPB - 1) / TPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * TPB + wave / KS, ksh = wave % KS;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  bf16x8 qf[H * C::NK];
  load_stationary<H, DH>(qf, q + ((long long)b * N + qrow) * C::D, g4);
  float lse[H];
#pragma unroll
  for (int h = 0; h < H; ++h) lse[h] = lse2[((long long)b * H + h) * N + qrow];
  // last feature block of a head: per-lane source of the transposing read (element offsets from Kc; spare lanes: the ones strip)
  const bool spare_lane = SPARE && (16 * (C::DT - 1) + 4 * (l15 & 3) >= DH);
  const int last_base = spare_lane ? (int)(ones - Kc) : (4 * g4 + (l15 >> 2)) * C::PITCH + 16 * (C::DT - 1) + 4 * (l15 & 3);
  // table: W / keep (the statistics are those of sum_h W[g,h] P~_h); accumulator start: -shift_g
  MixOp op;
  make_mix_op(op, W, nullptr, l15, g4, rng.inv_keep);
  f32x4 cin;
  {
    const int hh = g4 & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sw = 0.f;
#pragma unroll
      for (int h = 0; h < 8; ++h) sw += W[(4 * hh + j) * 8 + h];
      cin[j] = -sw / (float)N;
    }
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  // P k (the UN-dropped probabilities times the keys, per head): the backward's dq sweep needs sum_k P k to take delta out
  // of its element chain (dq = scale (sum_k P~ dP~ k - delta sum_k P k)), and this sweep has the probabilities and the
  // key chunk at hand with registers to spare
  f32x4 pacc[H][C::DT];                       // rows = features 16 dt + 4 g4 + r of head h, column = query l15
  // s_b = sum_k bf16(P): the row sum of the probabilities AS THE PRODUCT ABOVE SEES THEM (row_norm_note: the fused dq sweep
  // needs sum_k [bf16(u) - bf16(P) delta] = 0 to hold for the rounded operands, or saturated rows - true dS = 0 - keep
  // 2^-9 |u| |k| of noise each)
  float sb[H];
#pragma unroll
  for (int h = 0; h < H; ++h) sb[h] = 0.f;
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) pacc[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  const int nchunks = (ntiles + CK - 1) / CK;
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Kc;
  st_Kc.fetch(kb, min(CK, ntiles) * 16, tid);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    __syncthreads();                                  // every wave has finished with the previous chunk
    st_Kc.commit(Kc, nt * 16, tid);
    __syncthreads();
    {                                               // next chunk in flight during the tile loop (the last trip re-fetches its own)
      const int cn = ch + 1 < nchunks ? ch + 1 : ch;
      st_Kc.fetch(kb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
    }
    if (active)
      for (int kc = (KS == 1 ? 0 : ((ch * CK) & 1) ^ ksh); kc < nt; kc += KS) {
        if constexpr (PC) VU_IGLP_HINT(VU_IGLP_FW);
        f32x4 S[H];
        tile_logits<H, DH>(S, Kc, kc, qf, l15, g4);
        tag_probs<H>(S, lse, c, rng, wlane + 4u * (uint32_t)(ch * CK + kc), hstride);
        bf16x8 pk[4];
        pack_heads(S, pk);
        if constexpr (PC) pc_store(pk, const_cast<u32x4_t*>(pc_tile(pcache, b, ntiles, tq, ch * CK + kc, lane)));
        const int last_kc = last_base + (spare_lane ? 0 : kc * 16 * C::PITCH);
#pragma unroll
        for (int h = 0; h < H; ++h) {                  // a logits-shaped tile is the B operand of a key-contracting product as it stands
          const f32x4 p4 = {fabsf(S[h][0]), fabsf(S[h][1]), fabsf(S[h][2]), fabsf(S[h][3])};
          const s16x4 bop = pack4s(p4);
          if constexpr (!SPARE) {
            const u32x2_t bw = __builtin_bit_cast(u32x2_t, bop);
            sb[h] = sum2_bf16(bw[1], sum2_bf16(bw[0], sb[h]));
          }
#pragma unroll
          for (int dt = 0; dt < C::DT - 1; ++dt)
            pacc[h][dt] = mfma16(tr_operand<C::PITCH>(Kc, kc * 16, h * DH + 16 * dt, l15, g4), bop, pacc[h][dt]);
          pacc[h][C::DT - 1] = mfma16(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Kc + last_kc + h * DH)), bop, pacc[h][C::DT - 1]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pk[r] = relu_packed(pk[r]);
        f32x4 A[2][4];
        mix_ml(A, op, pk, cin);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += A[half][r][j]; s2[j] = fmaf(A[half][r][j], A[half][r][j], s2[j]); }
      }
  }
  if constexpr (KS == 2) {                  // (pair_park / pair_take: the odd wave's P k partial and row sums into the even wave's)
    __syncthreads();
    float* cs = reinterpret_cast<float*>(smem_raw) + (wave >> 1) * ((H * C::DT + 2) * 256);
    f32x4 sb4[2] = {f32x4{sb[0], sb[1], sb[2], sb[3]}, f32x4{sb[4], sb[5], sb[6], sb[7]}};
    if (ksh) {
#pragma unroll
      for (int h = 0; h < H; ++h) pair_park(pacc[h], cs + h * C::DT * 256, lane);
      pair_park(sb4, cs + H * C::DT * 256, lane);
    }
    __syncthreads();
    if (!ksh) {
#pragma unroll
      for (int h = 0; h < H; ++h) pair_take(pacc[h], cs + h * C::DT * 256, lane);
      pair_take(sb4, cs + H * C::DT * 256, lane);
#pragma unroll
      for (int h = 0; h < H; ++h) sb[h] = sb4[h >> 2][h & 3];
    }
  }
  const bool writer = active && ksh == 0;
  if (writer) {
    float* prow = pk_out + ((long long)b * N + qrow) * C::D;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int f = 16 * dt + 4 * g4;
        if (f < DH) *reinterpret_cast<f32x4*>(prow + h * DH + f) = pacc[h][dt];
      }
  }
  if constexpr (SPARE) {              // the spare rows of the last feature block (lane group 2, register 0) hold sum_k bf16(P)
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (writer && g4 == 2) rinv[((long long)b * H + h) * N + qrow] = 1.0f / pacc[h][C::DT - 1][0];
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h) {     // the four lane groups of a query hold its four key quarters
      float x = sb[h];
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      if (writer && g4 == 0) rinv[((long long)b * H + h) * N + qrow] = 1.0f / x;
    }
  }
  // lanes with the same head half (g4 & 1) hold the same 4 heads: reduce over q (16 lanes) and over a (g4 >> 1)
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a = active ? s1[j] : 0.f, b2 = active ? s2[j] : 0.f;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b2 += __shfl_xor(b2, m, 64); }
    a += __shfl_xor(a, 32, 64); b2 += __shfl_xor(b2, 32, 64);
    if (l15 == 0 && g4 < 2) { red[wave * 16 + 4 * g4 + j] = a; red[wave * 16 + 8 + 4 * g4 + j] = b2; }
  }
  __syncthreads();
  if (tid < 16) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) a += red[w * 16 + tid];
    partials[(long long)blockIdx.x * 16 + tid] = a;        // [0..7] sums, [8..15] sums of squares (shifted)
  }
}

// ---- apply pass --------------------------------------------------------------------------------------------------
// PC: the sign-tagged probabilities come from the cache the moments sweep wrote (one 4 KB tile per step, the next one in flight
// during the mix / PV products of the current one): no q, no K chunk, no logits, exp2, hash or pack
template <int DH, int WPB, int CK, int KS = 1, bool PC = false>
__global__ __launch_bounds__(WPB * 64, 2) void flash2_apply_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const float* __restrict__ lse2,
    const float* __restrict__ stats, bf16_t* __restrict__ O, int B, int N, float c, vu_rng rng_in, const void* __restrict__ pcache) {
  constexpr int H = 8, FB = DH / 8;
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Vc = PC ? Kc : Kc + CK * 16 * C::PITCH;
  bf16_t* Zr = Vc + CK * 16 * C::PITCH;                                   // zero region for the dead operand halves
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  constexpr int TPB = WPB / KS;
  const int ntiles = N >> 4, per = (ntiles + TPB - 1) / TPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * TPB + wave / KS, ksh = wave % KS;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  const bf16_t* vb = v + (long long)b * N * C::D;
  bf16x8 qf[H * C::NK];
  float lse[H];
  if constexpr (!PC) {
    load_stationary<H, DH>(qf, q + ((long long)b * N + qrow) * C::D, g4);
#pragma unroll
    for (int h = 0; h < H; ++h) lse[h] = lse2[((long long)b * H + h) * N + qrow];
  }
  // PC: the wave's next tile(s) of row tq (it takes tiles ksh, ksh + KS, ...).  Unsplit sweeps keep TWO tiles in flight (the sweep is
  // bound by the cache stream: 4 KB per tile and wave), even tiles in nx, odd ones in nx1 (every chunk starts at an even tile)
  constexpr int RD = VU_PC_RING;
  PRing<RD> ring;
  if constexpr (PC) ring.init(reinterpret_cast<unsigned char*>(Zr + tr_zero_elems<H, DH>()), wave, lane);
  for (int i = tid; i < tr_zero_elems<H, DH>() / 8; i += WPB * 64) *reinterpret_cast<uint4*>(Zr + i * 8) = make_uint4(0, 0, 0, 0);
  MixOp op;                                                               // gamma rstd W / keep
  make_mix_op(op, stats + VU_BN_STATS_FWK(H), nullptr, l15, g4);
  f32x4 cin;                                                              // folded bias of the lane's 4 heads
#pragma unroll
  for (int j = 0; j < 4; ++j) cin[j] = stats[H * H + 4 * (g4 & 1) + j];
  TrSrc<H, DH> src;
  src.init(l15, g4, (int)(Zr - Vc));
  f32x4 oacc[4][FB];          // [j][feature block]: rows (head half, 8 features) x 16 queries
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) oacc[j][fb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  const int nchunks = (ntiles + CK - 1) / CK;
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Kc;
  if constexpr (!PC) st_Kc.fetch(kb, min(CK, ntiles) * 16, tid);
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Vc;
  st_Vc.fetch(vb, min(CK, ntiles) * 16, tid);
  const long long trow = ((long long)b * ntiles + tq) * ntiles;      // PC: first tile of the wave's row of the cache
  if constexpr (PC) {
    if (active) {
#pragma unroll
      for (int i = 0; i < RD; ++i) ring.issue(pcache, trow + min(ksh + i * KS, ntiles - 1), i);
    }
  }
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    if (!(PC && (VU_PC_ABLATE & 2) && ch > 0)) {
    __syncthreads();                                  // every wave has finished with the previous chunk
    if constexpr (!PC) st_Kc.commit(Kc, nt * 16, tid);
    st_Vc.commit(Vc, nt * 16, tid);
    __syncthreads();
    {                                               // next chunk in flight during the tile loop (the last trip re-fetches its own)
      const int cn = ch + 1 < nchunks ? ch + 1 : ch;
      if constexpr (!PC) st_Kc.fetch(kb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
      st_Vc.fetch(vb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
    }
    }
    if (active)
      for (int kc = (KS == 1 ? 0 : ((ch * CK) & 1) ^ ksh); kc < nt; kc += KS) {
        bf16x8 pk[4];
        int slot = 0;
        if constexpr (PC) {
          VU_IGLP_HINT(VU_IGLP_FW);
          slot = ring.take(pk, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[r] = relu_packed(pk[r]);         // P~ (dropped = 0)
        } else {
          f32x4 S[H];
          tile_logits<H, DH>(S, Kc, kc, qf, l15, g4);
          tag_probs<H, true>(S, lse, c, rng, wlane + 4u * (uint32_t)(ch * CK + kc), hstride);      // P~ (dropped = 0)
          pack_heads(S, pk);
        }
        f32x4 A[2][4];
        mix_ml(A, op, pk, cin);                       // A^ in ML
        if constexpr (PC) ring.refill(pcache, trow + min(ch * CK + kc + RD * KS, ntiles - 1), slot);
        const int tb = src.base(kc);
        if (PC && (VU_PC_ABLATE & 4)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) oacc[j][0] += A[0][j] + A[1][j];
          continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                // the two key halves of the tile as the two k-blocks of one K = 32 product (join_k)
          const f32x4 a0 = {A[0][0][j], A[0][1][j], A[0][2][j], A[0][3][j]}, a1 = {A[1][0][j], A[1][1][j], A[1][2][j], A[1][3][j]};
          const bf16x8 b8 = join_k(pack4s(a0), pack4s(a1));
#pragma unroll
          for (int fb = 0; fb < FB; ++fb)
            oacc[j][fb] = mfma32(join_k(tr_half<H, DH>(Vc, tb, 0, j, fb), tr_half<H, DH>(Vc, tb, 1, j, fb)), b8, oacc[j][fb]);
        }
      }
  }
  if constexpr (PC) ring.drain();           // (the clamped refills behind the last tiles: nothing may land in LDS after the wave has left)
  if constexpr (KS == 2) {
    __syncthreads();
    float* cs = reinterpret_cast<float*>(smem_raw) + (wave >> 1) * (4 * FB * 256);
    if (ksh) {
#pragma unroll
      for (int j = 0; j < 4; ++j) pair_park(oacc[j], cs + j * FB * 256, lane);
    }
    __syncthreads();
    if (!ksh) {
#pragma unroll
      for (int j = 0; j < 4; ++j) pair_take(oacc[j], cs + j * FB * 256, lane);
    }
  }
  if (active && ksh == 0) {   // accumulator row 4 g4 + jj: head 4 (g4 >> 1) + j, feature 8 fb + 4 (g4 & 1) + jj
    bf16_t* orow = O + ((long long)b * N + qrow) * C::D;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb)
        *reinterpret_cast<bf16x4*>(orow + (4 * (g4 >> 1) + j) * DH + 8 * fb + 4 * (g4 & 1)) = pack4(oacc[j][fb]);
  }
}

// ---- backward: shared pieces ------------------------------------------------------------------------------------
struct Bwd2Ops {
  MixOp xk2;          // -m2_g rstd_g W[g,h] / keep
  bf16x8 id[2];
  BackOp back;        // gamma_g rstd_g W[g,h] / keep, transposed use
  f32x4 cin;          // -m1_g - m2_g Xc_g of the lane's 4 heads
};
__device__ __forceinline__ void make_bwd2_ops(Bwd2Ops& o, const float* __restrict__ stats, int l15, int g4) {
  constexpr int H = 8;
  make_mix_op(o.xk2, stats + VU_BN_STATS_XK(H), stats + H * H + 4 * H, l15, g4, -1.f);      // rows scaled by -m2_g
  make_identity_op(o.id, l15, g4);
  make_back_op(o.back, stats + VU_BN_STATS_FWK(H), l15, g4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = 4 * (g4 & 1) + j;
    o.cin[j] = -stats[H * H + 3 * H + g] - stats[H * H + 4 * H + g] * stats[2 * H * H + 5 * H + g];
  }
}
// S (logits, OL) -> tagML (sign-tagged probabilities, ML); then dA^ (OL, computed here, late, so that its 32 registers
// are not live beside the logits) -> e (ML).  STREAM_A as in tile_prod; Dc / dst: streaming chunk and stationary image of
// the dA^ product (V rows and dO in the q-major sweeps, dO rows and V in the key-major one).
// KEPT_ONLY: T receives the kept probabilities P~ (dropped entries 0) instead of the signed tags - for sweeps that never
// need the un-dropped |P| (the fused dq sweep): saves their per-element max(tag, 0).
template <int H, int DH, bool STREAM_A, bool KEPT_ONLY = false>
__device__ __forceinline__ void bwd2_chain(f32x4 (&S)[H], const bf16_t* Dc, int kc, const bf16_t* dst, const float (&lse)[H], float c,
                                           const vu_rng& rng, uint32_t wt, uint32_t hstride, const Bwd2Ops& o, f32x4 (&T)[2][4],
                                           f32x4 (&E)[2][4], int l15, int g4) {
  tag_probs<H, KEPT_ONLY>(S, lse, c, rng, wt, hstride);
  bf16x8 pk[4];
  pack_heads(S, pk);
  convert_ml(T, o.id, pk, false);                        // tags (or, KEPT_ONLY, P~) in ML: exact, already bf16
  if constexpr (!KEPT_ONLY) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = relu_packed(pk[r]);
  }
  mix_ml(E, o.xk2, pk, o.cin);                           // -m1 - m2 x^
  LDS_FENCE();
  f32x4 Dh[H];
  tile_prod<H, DH, STREAM_A>(Dh, Dc, kc, nullptr, dst, nullptr, l15, g4);
  pack_heads(Dh, pk);
  convert_ml(E, o.id, pk, true);                         // + dA^  = e
}
// The same chain behind cached probabilities (PC sweeps), in two pieces so that the caller can issue the load of its next cache tile
// between them (after the last use of the registers the tile arrived in):
//   head: pk = the cached tags (KEPT_ONLY: already max(tag, 0) = P~) -> T (ML); pk <- P~
//   tail: E = -m1 - m2 x^ + dA^ (dA^ formed here from the streamed chunk and the stationary image)
template <bool KEPT_ONLY>
__device__ __forceinline__ void bwd2_head_pk(bf16x8 (&pk)[4], const Bwd2Ops& o, f32x4 (&T)[2][4]) {
  convert_ml(T, o.id, pk, false);
  if constexpr (!KEPT_ONLY) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = relu_packed(pk[r]);
  }
}
template <int H, int DH, bool STREAM_A>
__device__ __forceinline__ void bwd2_tail_pk(bf16x8 (&pk)[4], const bf16_t* Dc, int kc, const bf16_t* dst, const Bwd2Ops& o, f32x4 (&E)[2][4],
                                             int l15, int g4) {
  mix_ml(E, o.xk2, pk, o.cin);                           // -m1 - m2 x^
  LDS_FENCE();
  f32x4 Dh[H];
  tile_prod<H, DH, STREAM_A>(Dh, Dc, kc, nullptr, dst, nullptr, l15, g4);
  pack_heads(Dh, pk);
  convert_ml(E, o.id, pk, true);                         // + dA^  = e
}
// dP (ML) of one (half, r) register group: transposed mix of e
// (e as a bf16 hi + lo pair: e is what is left of dA^ after the BatchNorm projections, and the q / k gradients downstream
// are sums of ~1e6 cancelling terms - with a single bf16 rounding of e the conv-weight gradients missed the 5e-2 tolerance)
__device__ __forceinline__ f32x4 bwd2_dp(const f32x4& e, const BackOp& bk) {
  const s16x4 b = pack4s(e);
  const s16x4 bl = pack4s(residual4(e, b));
  // (round 3: hi b + hi bl as one K = 32 product (hi | hi)(b | bl) followed by the K = 16 product lo b gave WRONG results - an
  // accumulator chained from a 16x16x32 into a 16x16x16 instruction; with lo b as the K = 32 product (lo | 0)(b | b) the
  // results are right and the sweeps slower - dk 367 -> 373 us, dq+delta 408 -> 416: the operand copies cost more than the
  // eight matrix instructions per tile return.  Three K = 16 products stay.)
  f32x4 dp = mfma16(bk.hi, b, f32x4{0.f, 0.f, 0.f, 0.f});
  dp = mfma16(bk.lo, b, dp);
  return mfma16(bk.hi, bl, dp);
}

// ---- sweep 1: delta and the head-mix gradient sums -------------------------------------------------------------
template <int DH, int WPB, int CK>
__global__ __launch_bounds__(WPB * 64, 2) void flash2_bwd_delta_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse2, const float* __restrict__ rinv, const float* __restrict__ stats, float* __restrict__ delta,
    float* __restrict__ partials, int B, int N, float c, vu_rng rng_in) {
  constexpr int H = 8, NT = H * H + H;
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Vc = Kc + CK * 16 * C::PITCH;
  bf16_t* Qs = Vc + CK * 16 * C::PITCH + (threadIdx.x >> 6) * 32 * C::PITCH;
  bf16_t* dOs = Qs + 16 * C::PITCH;
  bf16_t* img = Vc + CK * 16 * C::PITCH + WPB * 32 * C::PITCH + (threadIdx.x >> 6) * 640;      // [32 positions][20: 16 + pad]
  float* red = reinterpret_cast<float*>(Kc);                              // [WPB][NT], after the last tile (aliases the K chunk)
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int ntiles = N >> 4, per = (ntiles + WPB - 1) / WPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * WPB + wave;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  const bf16_t* vb = v + (long long)b * N * C::D;
  zero_pads<H, DH>(Vc + CK * 16 * C::PITCH, WPB * 32, tid, WPB * 64);       // (tile_prod reads the pads of the stationary images as zeros)
  stage_own_rows<H, DH>(Qs, q + ((long long)b * N + tq * 16) * C::D, lane);
  stage_own_rows<H, DH>(dOs, dO + ((long long)b * N + tq * 16) * C::D, lane);
  float lse[H];
#pragma unroll
  for (int h = 0; h < H; ++h) lse[h] = lse2[((long long)b * H + h) * N + qrow];
  Bwd2Ops ops;
  make_bwd2_ops(ops, stats, l15, g4);
  f32x4 dl = {0.f, 0.f, 0.f, 0.f}, sp = {0.f, 0.f, 0.f, 0.f}, tc = {0.f, 0.f, 0.f, 0.f}, Tacc = {0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  const int nchunks = (ntiles + CK - 1) / CK;
  const int hh = g4 & 1, a2 = g4 >> 1;
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Kc;
  st_Kc.fetch(kb, min(CK, ntiles) * 16, tid);
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Vc;
  st_Vc.fetch(vb, min(CK, ntiles) * 16, tid);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    __syncthreads();                                  // every wave has finished with the previous chunk
    st_Kc.commit(Kc, nt * 16, tid);
    st_Vc.commit(Vc, nt * 16, tid);
    __syncthreads();
    {                                               // next chunk in flight during the tile loop (the last trip re-fetches its own)
      const int cn = ch + 1 < nchunks ? ch + 1 : ch;
      st_Kc.fetch(kb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
      st_Vc.fetch(vb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
    }
    if (active)
      for (int kc = 0; kc < nt; ++kc) {
        f32x4 S[H], T[2][4], E[2][4];
        tile_prod<H, DH, true>(S, Kc, kc, nullptr, Qs, nullptr, l15, g4);
        bwd2_chain<H, DH, true>(S, Vc, kc, dOs, lse, c, rng, wlane + 4u * (uint32_t)(ch * CK + kc), hstride, ops, T, E, l15, g4);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 dp = bwd2_dp(E[half][r], ops.back);
            f32x4 ph;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              ph[j] = fmaxf(T[half][r][j], 0.f);
              dl[j] = fmaf(ph[j], dp[j], dl[j]);
              sp[j] += fabsf(T[half][r][j]);        // row sum of the probabilities as this chain holds them (bf16 values)
              tc[j] += E[half][r][j];
            }
            // sum over positions of e_g P^_h as X^T X: image row = position (q, a) of this (half, r), columns
            // [e_0..7 | P^_0..7]; the lane writes its head half of both.  P^ is exact in bf16 here (it went through the bf16
            // conversion to ML); e is written as a bf16 hi + lo pair in two rounds (e sums to ~0 over the positions while
            // P^ is nearly constant, so plain bf16 rounding of e does not cancel: measured 6 - 20 % error on dW).
            bf16_t* row = img + (a2 * 16 + l15) * 20 + 4 * hh;
            const s16x4 ehi = pack4s(E[half][r]);
            const f32x4 elo4 = residual4(E[half][r], ehi);
            *reinterpret_cast<s16x4*>(row) = ehi;
            *reinterpret_cast<s16x4*>(row + 8) = pack4s(ph);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {       // 32 positions = 2 k-blocks of 16
              const s16x4 x = tr_operand<20>(img, 16 * pb, 0, l15, g4);
              Tacc = mfma16(x, x, Tacc);
            }
            *reinterpret_cast<s16x4*>(row) = pack4s(elo4);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
              const s16x4 x = tr_operand<20>(img, 16 * pb, 0, l15, g4);
              Tacc = mfma16(x, x, Tacc);
            }
          }
      }
  }
  // delta: the lanes (q, hh) and (q, hh + 2) hold the two key groups of the same 4 heads
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float d = dl[j], sx = sp[j];
    d += __shfl_xor(d, 32, 64);
    sx += __shfl_xor(sx, 32, 64);
    if (active && g4 < 2) delta[((long long)b * H + 4 * hh + j) * N + qrow] = d / sx;   // row_norm_note
  }
  // T: accumulator rows c = 4 g4 + jj, column c' = l15: wanted rows 0..7 (e_g), columns 8..15 (P^_h)
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int g = 4 * g4 + jj;
    if (g < 8 && l15 >= 8) red[wave * NT + g * H + (l15 - 8)] = active ? Tacc[jj] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = active ? tc[j] : 0.f;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    x += __shfl_xor(x, 32, 64);
    if (l15 == 0 && g4 < 2) red[wave * NT + H * H + 4 * g4 + j] = x;
  }
  __syncthreads();
  for (int i = tid; i < NT; i += WPB * 64) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) a += red[w * NT + i];
    partials[(long long)blockIdx.x * NT + i] = a;
  }
}

// ---- sweep 2: dq ---------------------------------------------------------------------------------------------------
template <int DH, int WPB, int CK>
__global__ __launch_bounds__(WPB * 64, 2) void flash2_bwd_dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse2, const float* __restrict__ delta, const float* __restrict__ stats, bf16_t* __restrict__ dq,
    int B, int N, float c, float scale, vu_rng rng_in) {
  constexpr int H = 8, FB = DH / 8;
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Vc = Kc + CK * 16 * C::PITCH;
  bf16_t* Qs = Vc + CK * 16 * C::PITCH + (threadIdx.x >> 6) * 32 * C::PITCH;
  bf16_t* dOs = Qs + 16 * C::PITCH;
  bf16_t* Zr = Vc + CK * 16 * C::PITCH + WPB * 32 * C::PITCH;
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const int ntiles = N >> 4, per = (ntiles + WPB - 1) / WPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * WPB + wave;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  const bf16_t* vb = v + (long long)b * N * C::D;
  zero_pads<H, DH>(Vc + CK * 16 * C::PITCH, WPB * 32, tid, WPB * 64);       // (tile_prod reads the pads of the stationary images as zeros)
  stage_own_rows<H, DH>(Qs, q + ((long long)b * N + tq * 16) * C::D, lane);
  stage_own_rows<H, DH>(dOs, dO + ((long long)b * N + tq * 16) * C::D, lane);
  for (int i = tid; i < tr_zero_elems<H, DH>() / 8; i += WPB * 64) *reinterpret_cast<uint4*>(Zr + i * 8) = make_uint4(0, 0, 0, 0);
  float lse[H];
#pragma unroll
  for (int h = 0; h < H; ++h) lse[h] = lse2[((long long)b * H + h) * N + qrow];
  f32x4 dlt;
#pragma unroll
  for (int j = 0; j < 4; ++j) dlt[j] = delta[((long long)b * H + 4 * (g4 & 1) + j) * N + qrow];
  Bwd2Ops ops;
  make_bwd2_ops(ops, stats, l15, g4);
  TrSrc<H, DH> src;
  src.init(l15, g4, (int)(Zr - Kc));
  f32x4 acc[4][FB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) acc[j][fb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  const int nchunks = (ntiles + CK - 1) / CK;
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Kc;
  st_Kc.fetch(kb, min(CK, ntiles) * 16, tid);
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Vc;
  st_Vc.fetch(vb, min(CK, ntiles) * 16, tid);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    __syncthreads();                                  // every wave has finished with the previous chunk
    st_Kc.commit(Kc, nt * 16, tid);
    st_Vc.commit(Vc, nt * 16, tid);
    __syncthreads();
    {                                               // next chunk in flight during the tile loop (the last trip re-fetches its own)
      const int cn = ch + 1 < nchunks ? ch + 1 : ch;
      st_Kc.fetch(kb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
      st_Vc.fetch(vb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
    }
    if (active)
      for (int kc = 0; kc < nt; ++kc) {
        f32x4 S[H], T[2][4], E[2][4];
        tile_prod<H, DH, true>(S, Kc, kc, nullptr, Qs, nullptr, l15, g4);
        bwd2_chain<H, DH, true>(S, Vc, kc, dOs, lse, c, rng, wlane + 4u * (uint32_t)(ch * CK + kc), hstride, ops, T, E, l15, g4);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 dp = bwd2_dp(E[half][r], ops.back);
#pragma unroll
            for (int j = 0; j < 4; ++j) T[half][r][j] = fmaf(fmaxf(T[half][r][j], 0.f), dp[j], -fabsf(T[half][r][j]) * dlt[j]);   // dS
          }
        const int tb = src.base(kc);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const f32x4 d4 = {T[half][0][j], T[half][1][j], T[half][2][j], T[half][3][j]};
            const s16x4 bop = pack4s(d4);
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) acc[j][fb] = mfma16(tr_half<H, DH>(Kc, tb, half, j, fb), bop, acc[j][fb]);
          }
      }
  }
  if (active) {
    bf16_t* orow = dq + ((long long)b * N + qrow) * C::D;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        f32x4 o = acc[j][fb];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= scale;
        *reinterpret_cast<bf16x4*>(orow + (4 * (g4 >> 1) + j) * DH + 8 * fb + 4 * (g4 & 1)) = pack4(o);
      }
  }
}

// ---- sweeps 1 + 2 in one: delta, the head-mix gradient sums and dq ------------------------------------------------
// dS = P~ dP~ - P delta with delta = sum_k P~ dP~ known only after the whole row: instead of a sweep for delta and a
// second one for dq (each recomputing the chain), accumulate U = sum_k (P~ dP~) k here and take V = sum_k P k from the
// forward (flash2_moments_kernel): dq = scale (U - delta V).  The u = P~ dP~ terms go to the matrix cores as single bf16
// values like dS did; U - delta V is the covariance form of the same sum.
// PC: probabilities from the cache (see flash2_apply_kernel): no q image, no logits; the wave keeps only its dO rows
template <int DH, int WPB, int CK, int KS = 1, bool PC = false>
__global__ __launch_bounds__(WPB * 64, 2) void flash2_bwd_dqx_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse2, const float* __restrict__ rinv, const float* __restrict__ pkv, const float* __restrict__ stats,
    bf16_t* __restrict__ dq, float* __restrict__ delta, float* __restrict__ partials, int B, int N, float c, float scale,
    vu_rng rng_in, int want_dc, const void* __restrict__ pcache) {
  constexpr int H = 8, FB = DH / 8, NT = H * H + H, IMP = 16;
  constexpr int SR = PC ? 16 : 32;                  // stationary rows per wave: (q and) dO
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Kc = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Vc = Kc + CK * 16 * C::PITCH;
  bf16_t* Qs = Vc + CK * 16 * C::PITCH + (threadIdx.x >> 6) * SR * C::PITCH;
  bf16_t* dOs = PC ? Qs : Qs + 16 * C::PITCH;
  // [32 positions][16: e_0..7 | P^_0..7], 8-byte column blocks XOR-swizzled with (row >> 2) & 3 (pitch 32 B: unswizzled,
  // rows 4 apart would meet on the same banks in the stores)
  // (round 5, measured and not kept: eight images per wave in the cached form - the e_hi | P~ and e_lo | P~ images of four (half, r)
  // groups at a time, so that a tile's sixteen store -> transposing read -> product round trips become two batches: 341 - 348 us
  // against 339 - 342 at 64 images, 347 against 319 under the scheduling hint; the section is not bound by its LDS round trips)
  constexpr int NIM = 1;
  bf16_t* img = Vc + CK * 16 * C::PITCH + WPB * SR * C::PITCH + (threadIdx.x >> 6) * (32 * IMP * NIM);
  bf16_t* Zr = Vc + CK * 16 * C::PITCH + WPB * SR * C::PITCH + WPB * (32 * IMP * NIM);
  float* red = reinterpret_cast<float*>(Kc);                              // [WPB][NT], after the last tile (aliases the K chunk)
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  constexpr int TPB = WPB / KS;
  const int ntiles = N >> 4, per = (ntiles + TPB - 1) / TPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * TPB + wave / KS, ksh = wave % KS;
  const bool active = t < ntiles;
  const int tq = active ? t : ntiles - 1;
  const int qrow = tq * 16 + l15;
  const bf16_t* kb = k + (long long)b * N * C::D;
  const bf16_t* vb = v + (long long)b * N * C::D;
  zero_pads<H, DH>(Vc + CK * 16 * C::PITCH, WPB * SR, tid, WPB * 64);       // (tile_prod reads the pads of the stationary images as zeros)
  if constexpr (!PC) stage_own_rows<H, DH>(Qs, q + ((long long)b * N + tq * 16) * C::D, lane);
  stage_own_rows<H, DH>(dOs, dO + ((long long)b * N + tq * 16) * C::D, lane);
  for (int i = tid; i < tr_strip_elems<DH>() / 8; i += WPB * 64) *reinterpret_cast<uint4*>(Zr + i * 8) = make_uint4(0, 0, 0, 0);
  float lse[H];
  if constexpr (!PC) {
#pragma unroll
    for (int h = 0; h < H; ++h) lse[h] = lse2[((long long)b * H + h) * N + qrow];
  }
  PTile nx;
  if constexpr (PC) pc_load(nx, pc_tile(pcache, b, ntiles, tq, min(ksh, ntiles - 1), lane));
  Bwd2Ops ops;
  make_bwd2_ops(ops, stats, l15, g4);
  TrSrc2<H, DH> src;
  src.init(l15, g4, (int)(Zr - Kc));
  f32x4 acc[4][FB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) acc[j][fb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dl: sum_k u (fp32 products, what the dk sweep's dS = P~ dP~ - P delta cancels against); dlb: sum_k bf16(u), the operand
  // values of THIS sweep's dq product (row_norm_note)
  f32x4 dl = {0.f, 0.f, 0.f, 0.f}, dlb = {0.f, 0.f, 0.f, 0.f}, tc = {0.f, 0.f, 0.f, 0.f}, Tacc = {0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);
  const uint32_t wlane = (uint32_t)((((unsigned long long)b * N + qrow) * (unsigned long long)N) >> 2) + (uint32_t)g4;
  const int nchunks = (ntiles + CK - 1) / CK;
  const int hh = g4 & 1, a2 = g4 >> 1;
  // image addresses: the lane's row (a2 16 + l15) and its two swizzled column blocks; the transposed reads of k-block pb
  const int sw = (l15 >> 2) & 3;
  bf16_t* im_e = img + (a2 * 16 + l15) * IMP + 4 * (hh ^ sw);
  bf16_t* im_p = img + (a2 * 16 + l15) * IMP + 4 * ((2 + hh) ^ sw);
  const bf16_t* im_r = img + (4 * g4 + (l15 >> 2)) * IMP + 4 * ((l15 & 3) ^ g4);
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Kc;
  st_Kc.fetch(kb, min(CK, ntiles) * 16, tid);
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Vc;
  st_Vc.fetch(vb, min(CK, ntiles) * 16, tid);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    if (!(PC && (VU_DQX_ABLATE & 8) && ch > 0)) {
    __syncthreads();                                  // every wave has finished with the previous chunk
    st_Kc.commit(Kc, nt * 16, tid);
    st_Vc.commit(Vc, nt * 16, tid);
    __syncthreads();
    {                                               // next chunk in flight during the tile loop (the last trip re-fetches its own)
      const int cn = ch + 1 < nchunks ? ch + 1 : ch;
      st_Kc.fetch(kb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
      st_Vc.fetch(vb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
    }
    }
    if (active)
      for (int kc = (KS == 1 ? 0 : ((ch * CK) & 1) ^ ksh); kc < nt; kc += KS) {
        f32x4 T[2][4], E[2][4];
        if constexpr (PC) {
          if constexpr (DH == 24 && KS == 1 && WPB == 4) VU_IGLP_HINT(VU_IGLP_DQ);      // (the other instantiations measured slower with it)
          bf16x8 pk[4];
          pc_take(pk, nx);
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[r] = relu_packed(pk[r]);         // the kept probabilities P~
          pc_load(nx, pc_tile(pcache, b, ntiles, tq, min(ch * CK + kc + KS, ntiles - 1), lane));
          bwd2_head_pk<true>(pk, ops, T);
          if (VU_DQX_ABLATE & 16) {                      // (measurement build: no dA^ product, no mix)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
              for (int r = 0; r < 4; ++r) E[half][r] = T[half][r] + ops.cin;
          } else
          bwd2_tail_pk<H, DH, true>(pk, Vc, kc, dOs, ops, E, l15, g4);
        } else {
          f32x4 S[H];
          tile_prod<H, DH, true>(S, Kc, kc, nullptr, Qs, nullptr, l15, g4);
          bwd2_chain<H, DH, true, true>(S, Vc, kc, dOs, lse, c, rng, wlane + 4u * (uint32_t)(ch * CK + kc), hstride, ops, T, E, l15, g4);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 dp = (PC && (VU_DQX_ABLATE & 4)) ? E[half][r] : bwd2_dp(E[half][r], ops.back);
            f32x4 ph;
#pragma unroll
            for (int j = 0; j < 4; ++j) ph[j] = T[half][r][j];                 // (already the kept probabilities: bwd2_chain<KEPT_ONLY>)
            // sum of e_g = the gradient of the mix bias: identically zero under batch statistics (BatchNorm removes any
            // constant shift of its input), so the training sweep skips it; with running statistics (eval) it is real
            if (want_dc) {
#pragma unroll
              for (int j = 0; j < 4; ++j) tc[j] += E[half][r][j];
            }
            // sum over positions of e_g P^_h as X^T X (see flash2_bwd_delta_kernel): e as a bf16 hi + lo pair in two rounds
            const s16x4 ehi = pack4s(E[half][r]);
            const f32x4 elo4 = residual4(E[half][r], ehi);
            if (PC && (VU_DQX_ABLATE & 1)) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { const float u = ph[j] * dp[j]; dl[j] += u + elo4[j]; T[half][r][j] = u; }
              continue;
            }
            *reinterpret_cast<s16x4*>(im_e) = ehi;
            *reinterpret_cast<s16x4*>(im_p) = pack4s(ph);
            {                                      // 32 positions = 2 k-blocks of 16 = one K = 32 product (join_k)
              const bf16x8 x = join_k(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)im_r),
                                      __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(im_r + 16 * IMP)));
              Tacc = mfma32(x, x, Tacc);
            }
            *reinterpret_cast<s16x4*>(im_e) = pack4s(elo4);
            {
              const bf16x8 x = join_k(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)im_r),
                                      __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(im_r + 16 * IMP)));
              Tacc = mfma32(x, x, Tacc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float u = ph[j] * dp[j];         // P~ dP~
              dl[j] += u;
              T[half][r][j] = u;
            }
          }
        if (PC && (VU_DQX_ABLATE & 2)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j][0] += f32x4{T[0][0][j], T[0][1][j], T[1][2][j], T[1][3][j]};
        } else
        {                                          // the two key halves of the tile as the two k-blocks of one K = 32 product (join_k)
          const int tb0 = src.base(kc, 0), tb1 = src.base(kc, 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s16x4 bop[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const f32x4 d4 = {T[half][0][j], T[half][1][j], T[half][2][j], T[half][3][j]};
              bop[half] = pack4s(d4);
              const u32x2_t bw = __builtin_bit_cast(u32x2_t, bop[half]);
              dlb[j] = sum2_bf16(bw[1], sum2_bf16(bw[0], dlb[j]));
            }
            const bf16x8 b8 = join_k(bop[0], bop[1]);
#pragma unroll
            for (int fb = 0; fb < FB; ++fb)
              acc[j][fb] = mfma32(join_k(tr_rel<DH>(Kc, tb0, j, fb), tr_rel<DH>(Kc, tb1, j, fb)), b8, acc[j][fb]);
          }
        }
      }
  }
  if constexpr (KS == 2) {                  // the odd wave's partial U, row sums and head-mix gradient sums into the even wave's
    __syncthreads();
    float* cs = reinterpret_cast<float*>(smem_raw) + (wave >> 1) * ((4 * FB + 4) * 256);
    f32x4 misc[4] = {dl, dlb, tc, Tacc};
    if (ksh) {
#pragma unroll
      for (int j = 0; j < 4; ++j) pair_park(acc[j], cs + j * FB * 256, lane);
      pair_park(misc, cs + 4 * FB * 256, lane);
    }
    __syncthreads();
    if (!ksh) {
#pragma unroll
      for (int j = 0; j < 4; ++j) pair_take(acc[j], cs + j * FB * 256, lane);
      pair_take(misc, cs + 4 * FB * 256, lane);
      dl = misc[0]; dlb = misc[1]; tc = misc[2]; Tacc = misc[3];
    }
    __syncthreads();                                  // (red below aliases the same LDS)
  }
  const bool writer = active && ksh == 0;
  // delta: the lanes (q, hh) and (q, hh + 2) hold the two key groups of the same 4 heads; the accumulators of this lane
  // belong to heads 4 (g4 >> 1) + j, whose delta sits in the lanes with hh = g4 >> 1
  f32x4 dsel;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float d = dl[j], db = dlb[j];
    d += __shfl_xor(d, 32, 64);
    db += __shfl_xor(db, 32, 64);
    const float rs = rinv[((long long)b * H + 4 * hh + j) * N + qrow];    // row_norm_note: 1 / sum_k bf16(P) (moments sweep)
    if (writer && g4 < 2) delta[((long long)b * H + 4 * hh + j) * N + qrow] = d * rs;
    dsel[j] = __shfl(db * rs, l15 + 16 * (g4 >> 1), 64);
  }
  if (writer) {
    bf16_t* orow = dq + ((long long)b * N + qrow) * C::D;
    const float* vrow = pkv + ((long long)b * N + qrow) * C::D;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const int off = (4 * (g4 >> 1) + j) * DH + 8 * fb + 4 * (g4 & 1);
        const f32x4 pv = *reinterpret_cast<const f32x4*>(vrow + off);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = scale * fmaf(-dsel[j], pv[r], acc[j][fb][r]);
        *reinterpret_cast<bf16x4*>(orow + off) = pack4(o);
      }
  }
  // T: accumulator rows c = 4 g4 + jj, column c' = l15: wanted rows 0..7 (e_g), columns 8..15 (P^_h)
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int g = 4 * g4 + jj;
    if (g < 8 && l15 >= 8) red[wave * NT + g * H + (l15 - 8)] = writer ? Tacc[jj] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = writer ? tc[j] : 0.f;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    x += __shfl_xor(x, 32, 64);
    if (l15 == 0 && g4 < 2) red[wave * NT + H * H + 4 * g4 + j] = x;
  }
  __syncthreads();
  for (int i = tid; i < NT; i += WPB * 64) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) a += red[w * NT + i];
    partials[(long long)blockIdx.x * NT + i] = a;
  }
}

// ---- sweeps 3 / 4: dk (DV = false) and dv (DV = true), key-major loop ------------------------------------------------
// PC: probabilities from the cache (column tk of the sample's tile grid): no key image, no logits; the dv sweep streams dO only
#ifndef VU_PC_DV_WAVES
#define VU_PC_DV_WAVES 2        // (cached form: dO chunk 25.6 KB + images 20 KB + tile ring 32 KB per workgroup: two workgroups per CU)
#endif
template <int DH, int WPB, int CK, bool DV, int KS = 1, bool PC = false>
__global__ __launch_bounds__(WPB * 64, DV ? (PC ? VU_PC_DV_WAVES : 3) : 2) void flash2_bwd_dkv_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse2, const float* __restrict__ delta, const float* __restrict__ stats, bf16_t* __restrict__ out,
    int B, int N, float c, float scale, vu_rng rng_in, const void* __restrict__ pcache) {
  constexpr int H = 8;
  constexpr int IMP = 20, IMS = 16 * IMP;         // image row pitch 40 B: the 16 rows of a store cover the 32 banks once
  typedef FC<H, DH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr bool NEEDQ = !(PC && DV);             // the q chunk: logits (recompute form) and the dk product
  bf16_t* Qc = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Dc = NEEDQ ? Qc + CK * 16 * C::PITCH : Qc;
  constexpr int SROWS = PC ? (DV ? 0 : 16) : (DV ? 16 : 32);      // stationary rows per wave: the keys (recompute form), and (dk only) their v rows
  bf16_t* Ks = Dc + CK * 16 * C::PITCH + (threadIdx.x >> 6) * SROWS * C::PITCH;
  bf16_t* Vs = PC ? Ks : Ks + 16 * C::PITCH;
  // wave-private transposition images, 2 x [16 q][IMP] per set.  PC: one set per head pair j (LDS is no longer scarce), so that a tile's
  // eight stores, eight transposing reads and sixteen products are three batches instead of four store -> read -> product round trips
  constexpr int IMGS = PC ? 4 : 1;
  bf16_t* img = Dc + CK * 16 * C::PITCH + WPB * SROWS * C::PITCH + (threadIdx.x >> 6) * (640 * IMGS);
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  constexpr int TPB = WPB / KS;
  const int ntiles = N >> 4, per = (ntiles + TPB - 1) / TPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * TPB + wave / KS, ksh = wave % KS;
  const bool active = t < ntiles;
  const int tk = active ? t : ntiles - 1;
  const bf16_t* qb = q + (long long)b * N * C::D;
  const bf16_t* dob = dO + (long long)b * N * C::D;
  if constexpr (!PC) stage_own_rows<H, DH>(Ks, k + ((long long)b * N + tk * 16) * C::D, lane);
  if (!DV) stage_own_rows<H, DH>(Vs, v + ((long long)b * N + tk * 16) * C::D, lane);
  zero_pads<H, DH>(Qc, (NEEDQ ? 2 : 1) * CK * 16 + WPB * SROWS, tid, WPB * 64);
  // PC: the tiles of column tk of the sample's grid.  dk (a long chain per tile): the next tile in registers; dv (little besides the
  // stream): the LDS-DMA ring (PRing)
  constexpr bool RING = PC && DV;
  constexpr int RD = VU_PC_RING;
  PTile nx;
  PRing<RD> ring;
  const long long tcol = (long long)b * ntiles * ntiles + tk;      // tile (b, qt, tk) = tcol + qt * ntiles
  if constexpr (RING) {
    ring.init(reinterpret_cast<unsigned char*>(Dc + CK * 16 * C::PITCH + WPB * SROWS * C::PITCH + WPB * (640 * IMGS)), wave, lane);
    if (active) {
#pragma unroll
      for (int i = 0; i < RD; ++i) ring.issue(pcache, tcol + (long long)min(ksh + i * KS, ntiles - 1) * ntiles, i);
    }
  } else if constexpr (PC) pc_load(nx, pc_tile(pcache, b, ntiles, min(ksh, ntiles - 1), tk, lane));
  Bwd2Ops ops;
  MixOp fw;
  f32x4 fcin = {0.f, 0.f, 0.f, 0.f};
  if (DV) {
    make_mix_op(fw, stats + VU_BN_STATS_FWK(H), nullptr, l15, g4);
#pragma unroll
    for (int j = 0; j < 4; ++j) fcin[j] = stats[H * H + 4 * (g4 & 1) + j];
  } else make_bwd2_ops(ops, stats, l15, g4);
  f32x4 oa[H][C::DT];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oa[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t hstride = (uint32_t)(((unsigned long long)N * N) >> 2);
  const uint32_t wkey = (uint32_t)(((unsigned long long)b * N * (unsigned long long)N) >> 2) + 4u * (uint32_t)tk + (uint32_t)g4;
  const uint32_t wq = (uint32_t)(N >> 2);
  const int hh = g4 & 1, a2 = g4 >> 1;
  const int nchunks = (ntiles + CK - 1) / CK;
  // register-staged prefetch of the next chunk: only the dv sweep has the registers for it (dk sits at the 256 cap)
  constexpr bool PRE = DV && CK > 1;
  ChunkStage<H, DH, CK * 16, WPB * 64> st_Qc, st_Dc;
  if constexpr (PRE) {
    if constexpr (NEEDQ) st_Qc.fetch(qb, min(CK, ntiles) * 16, tid);
    st_Dc.fetch(dob, min(CK, ntiles) * 16, tid);
  }
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nt = min(CK, ntiles - ch * CK);
    __syncthreads();
    if constexpr (PRE) {
      if constexpr (NEEDQ) st_Qc.commit(Qc, nt * 16, tid);
      st_Dc.commit(Dc, nt * 16, tid);
    } else {
      if constexpr (NEEDQ) load_chunk<H, DH>(Qc, qb + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
      load_chunk<H, DH>(Dc, dob + (long long)ch * CK * 16 * C::D, nt * 16, tid, WPB * 64);
    }
    __syncthreads();
    if constexpr (PRE) {
      const int cn = ch + 1 < nchunks ? ch + 1 : ch;
      if constexpr (NEEDQ) st_Qc.fetch(qb + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
      st_Dc.fetch(dob + (long long)cn * CK * 16 * C::D, min(CK, ntiles - cn * CK) * 16, tid);
    }
    if (active)
      for (int qc = (KS == 1 ? 0 : ((ch * CK) & 1) ^ ksh); qc < nt; qc += KS) {
        // row constants of the tile's queries (log-sum-exp of all heads, delta of the lane's 4 heads): L2-resident
        if constexpr (PC && !DV) VU_IGLP_HINT(VU_IGLP_DK);
        const long long qg = (long long)(ch * CK + qc) * 16 + l15;
        float lse[H];
        if constexpr (!PC) {
#pragma unroll
          for (int h = 0; h < H; ++h) lse[h] = lse2[((long long)b * H + h) * N + qg];
        }
        f32x4 dlt = {0.f, 0.f, 0.f, 0.f};
        if (!DV) {
#pragma unroll
          for (int j = 0; j < 4; ++j) dlt[j] = delta[((long long)b * H + 4 * hh + j) * N + qg];
        }
        f32x4 S[H], T[2][4];
        if constexpr (!PC) tile_prod<H, DH, false>(S, Qc, qc, nullptr, Ks, nullptr, l15, g4);
        const uint32_t wt = wkey + (uint32_t)((ch * CK + qc) * 16 + l15) * wq;
        if constexpr (DV) {
          bf16x8 pk[4];
          int slot = 0;
          if constexpr (PC) {
            slot = ring.take(pk, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) pk[r] = relu_packed(pk[r]);       // P~ (dropped = 0)
          } else {
            tag_probs<H, true>(S, lse, c, rng, wt, hstride);       // P~ (dropped = 0)
            pack_heads(S, pk);
          }
          mix_ml(T, fw, pk, fcin);                     // A^ (ML)
          if constexpr (PC) ring.refill(pcache, tcol + (long long)min(ch * CK + qc + RD * KS, ntiles - 1) * ntiles, slot);
        } else {
          f32x4 E[2][4];
          if constexpr (PC) {
            bf16x8 pk[4];
            pc_take(pk, nx);
            bwd2_head_pk<false>(pk, ops, T);           // T: the signed tags; pk: P~
            pc_load(nx, pc_tile(pcache, b, ntiles, min(ch * CK + qc + KS, ntiles - 1), tk, lane));
            bwd2_tail_pk<H, DH, false>(pk, Dc, qc, Vs, ops, E, l15, g4);
          } else
          bwd2_chain<H, DH, false>(S, Dc, qc, Vs, lse, c, rng, wt, hstride, ops, T, E, l15, g4);
#pragma unroll
          for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const f32x4 dp = bwd2_dp(E[half][r], ops.back);
#pragma unroll
              for (int j = 0; j < 4; ++j) T[half][r][j] = fmaf(fmaxf(T[half][r][j], 0.f), dp[j], -fabsf(T[half][r][j]) * dlt[j]);   // dS
            }
        }
        // contraction over the query (lane) index: per j the two heads j (lanes hh = 0) and 4 + j (hh = 1) go through
        // wave-private [query][key] images and come back transposed
        if constexpr (IMGS == 4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bf16_t* im = img + j * 640 + hh * IMS;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const f32x4 x4 = {T[half][0][j], T[half][1][j], T[half][2][j], T[half][3][j]};
              *reinterpret_cast<s16x4*>(im + l15 * IMP + 8 * half + 4 * a2) = pack4s(x4);
            }
          }
          s16x4 bop[4][2];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) bop[j][h2] = tr_operand<IMP>(img + j * 640 + h2 * IMS, 0, 0, l15, g4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const int h = 4 * h2 + j;
#pragma unroll
              for (int dt = 0; dt < C::DT; ++dt)
                oa[h][dt] = mfma16(tr_operand<C::PITCH>(DV ? Dc : Qc, qc * 16, h * DH + 16 * dt, l15, g4), bop[j][h2], oa[h][dt]);
            }
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16_t* im = img + hh * IMS;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const f32x4 x4 = {T[half][0][j], T[half][1][j], T[half][2][j], T[half][3][j]};
            *reinterpret_cast<s16x4*>(im + l15 * IMP + 8 * half + 4 * a2) = pack4s(x4);
          }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int h = 4 * h2 + j;
            const s16x4 bop = tr_operand<IMP>(img + h2 * IMS, 0, 0, l15, g4);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
              oa[h][dt] = mfma16(tr_operand<C::PITCH>(DV ? Dc : Qc, qc * 16, h * DH + 16 * dt, l15, g4), bop, oa[h][dt]);
          }
        }
        }
      }
  }
  if constexpr (RING) ring.drain();         // (the clamped refills behind the last tiles: nothing may land in LDS after the wave has left)
  if constexpr (KS == 2) {                  // (query split: the odd wave's partial dk / dv into the even wave's)
    __syncthreads();
    float* cs = reinterpret_cast<float*>(smem_raw) + (wave >> 1) * (H * C::DT * 256);
    if (ksh) {
#pragma unroll
      for (int h = 0; h < H; ++h) pair_park(oa[h], cs + h * C::DT * 256, lane);
    }
    __syncthreads();
    if (!ksh) {
#pragma unroll
      for (int h = 0; h < H; ++h) pair_take(oa[h], cs + h * C::DT * 256, lane);
    }
  }
  if (active && ksh == 0) {
    bf16_t* orow = out + ((long long)b * N + tk * 16 + l15) * C::D;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int f = 16 * dt + 4 * g4;
        if (f < DH) {
          f32x4 o = oa[h][dt];
          if (!DV) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= scale;
          }
          *reinterpret_cast<bf16x4*>(orow + h * DH + f) = pack4(o);
        }
      }
  }
}

// ---- cached dk sweep with LDS-DMA operand staging (round 5): Base / Large level 2 (d = 24), unsplit --------------------------------
// flash2_bwd_dkv_kernel<.., DV = false, PC> stages its q / dO chunks synchronously (load -> LDS between two barriers: at 256 registers
// it has none to prefetch into), so every second tile pays the L2 latency of a chunk.  Here the two streamed operands arrive by
// LDS-DMA into double-buffered 16-row images: the tile after the current one is in flight during the whole tile body, one barrier per
// tile, no staging registers.  An image [16][PITCH] is 6400 bytes; a DMA instruction writes 64 lanes x 16 bytes LINEARLY, so the
// padded image is filled by 8 pieces of 1 KiB whose per-lane SOURCE addresses skip nothing and clamp the pad vector of a row to its
// last data vector (the pads and the slack behind row 15 are never read as operands: the dA^ product re-reads a valid k-slot, the dk
// product's spare feature rows are not stored).  Same arithmetic in the same order as the flash2 kernel: bit-identical results.
__device__ __forceinline__ void pc_dma2(const void* tile, unsigned v0, unsigned v1, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
               "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(tile), "v"(v0), "v"(v1), "s"(lds_dst) : "memory", "scc");
}
template <int DH>
__global__ __launch_bounds__(256, 2) void flash3_dk_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                           const float* __restrict__ delta, const float* __restrict__ stats,
                                                           bf16_t* __restrict__ out, int B, int N, float scale, const void* __restrict__ pcache) {
  constexpr int H = 8, WPB = 4, IMP = 20, IMS = 16 * IMP, SLOT = 8192;
  typedef FC<H, DH> C;
  static_assert(16 * C::PITCH * 2 <= SLOT - 0 && (16 * C::PITCH * 2 + 1023) / 1024 <= 8, "a 16-row image must fit eight 1 KiB pieces");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* Qb = smem_raw;                                   // [2][SLOT]
  unsigned char* Db = smem_raw + 2 * SLOT;                        // [2][SLOT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  bf16_t* Vs = reinterpret_cast<bf16_t*>(smem_raw + 4 * SLOT) + wave * 16 * C::PITCH;
  bf16_t* img = reinterpret_cast<bf16_t*>(smem_raw + 4 * SLOT + WPB * 16 * C::PITCH * 2) + wave * (640 * 4);
  const int ntiles = N >> 4, per = (ntiles + WPB - 1) / WPB;
  int b, grp;
  work_item(blockIdx.x, B, per, b, grp);
  const int t = grp * WPB + wave;
  const bool active = t < ntiles;
  const int tk = active ? t : ntiles - 1;
  stage_own_rows<H, DH>(Vs, v + ((long long)b * N + tk * 16) * C::D, lane);
  for (int r = lane; r < 16; r += 64) *reinterpret_cast<uint4*>(Vs + r * C::PITCH + C::D) = make_uint4(0, 0, 0, 0);      // (tile_prod reads the stationary image's pads as zeros)
  // the wave's two pieces (2 wave, 2 wave + 1) of an image: per-lane byte offset of its source vector inside a 16-row tile
  constexpr int VPRP = C::PITCH / 8;                              // 16-byte vectors per image row, the pad vector included
  unsigned voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = (2 * wave + j) * 64 + lane;
    int row = i / VPRP, col = i - row * VPRP;
    if (row > 15) row = 15;
    if (col > C::VPR - 1) col = C::VPR - 1;
    voff[j] = (unsigned)(row * C::D * 2 + col * 16);
  }
  const unsigned ldsq = (unsigned)(size_t)Qb + (unsigned)(2 * wave) * 1024u, ldsd = (unsigned)(size_t)Db + (unsigned)(2 * wave) * 1024u;
  const char* qb = reinterpret_cast<const char*>(q + (long long)b * N * C::D);
  const char* dob = reinterpret_cast<const char*>(dO + (long long)b * N * C::D);
  auto issue = [&](int qt, int buf) {
    const unsigned long long aq = (unsigned long long)(qb + (long long)qt * 16 * C::D * 2), ad = (unsigned long long)(dob + (long long)qt * 16 * C::D * 2);
    // (readfirstlane returns a SIGNED int: each half goes through an unsigned before the halves are joined)
    const unsigned qlo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)aq), qhi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(aq >> 32));
    const unsigned dlo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)ad), dhi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ad >> 32));
    const void* pq = reinterpret_cast<const void*>(((unsigned long long)qhi << 32) | (unsigned long long)qlo);
    const void* pd = reinterpret_cast<const void*>(((unsigned long long)dhi << 32) | (unsigned long long)dlo);
    pc_dma2(pq, voff[0], voff[1], __builtin_amdgcn_readfirstlane(ldsq + (unsigned)buf * SLOT));
    pc_dma2(pd, voff[0], voff[1], __builtin_amdgcn_readfirstlane(ldsd + (unsigned)buf * SLOT));
  };
  issue(0, 0);
  PTile nx;
  pc_load(nx, pc_tile(pcache, b, ntiles, 0, tk, lane));
  Bwd2Ops ops;
  make_bwd2_ops(ops, stats, l15, g4);
  f32x4 oa[H][C::DT];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oa[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int hh = g4 & 1, a2 = g4 >> 1;
  for (int qt = 0; qt < ntiles; ++qt) {
    const int buf = qt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of tile qt (and its cache tile) have landed
    __syncthreads();                                            // ... everybody's; and every wave has finished with the other buffer
    if (qt + 1 < ntiles) issue(qt + 1, buf ^ 1);
    if (!active) continue;
    VU_IGLP_HINT(VU_IGLP_DK);
    const bf16_t* Qc = reinterpret_cast<const bf16_t*>(Qb + buf * SLOT);
    const bf16_t* Dc = reinterpret_cast<const bf16_t*>(Db + buf * SLOT);
    const long long qg = (long long)qt * 16 + l15;
    f32x4 dlt;
#pragma unroll
    for (int j = 0; j < 4; ++j) dlt[j] = delta[((long long)b * H + 4 * hh + j) * N + qg];
    f32x4 T[2][4], E[2][4];
    {
      bf16x8 pk[4];
      pc_take(pk, nx);
      bwd2_head_pk<false>(pk, ops, T);           // T: the signed tags; pk: P~
      pc_load(nx, pc_tile(pcache, b, ntiles, min(qt + 1, ntiles - 1), tk, lane));
      bwd2_tail_pk<H, DH, false>(pk, Dc, 0, Vs, ops, E, l15, g4);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 dp = bwd2_dp(E[half][r], ops.back);
#pragma unroll
        for (int j = 0; j < 4; ++j) T[half][r][j] = fmaf(fmaxf(T[half][r][j], 0.f), dp[j], -fabsf(T[half][r][j]) * dlt[j]);   // dS
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16_t* im = img + j * 640 + hh * IMS;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4 x4 = {T[half][0][j], T[half][1][j], T[half][2][j], T[half][3][j]};
        *reinterpret_cast<s16x4*>(im + l15 * IMP + 8 * half + 4 * a2) = pack4s(x4);
      }
    }
    s16x4 bop[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) bop[j][h2] = tr_operand<IMP>(img + j * 640 + h2 * IMS, 0, 0, l15, g4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int h = 4 * h2 + j;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) oa[h][dt] = mfma16(tr_operand<C::PITCH>(Qc, 0, h * DH + 16 * dt, l15, g4), bop[j][h2], oa[h][dt]);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (active) {
    bf16_t* orow = out + ((long long)b * N + tk * 16 + l15) * C::D;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int f = 16 * dt + 4 * g4;
        if (f < DH) {
          f32x4 o = oa[h][dt];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] *= scale;
          *reinterpret_cast<bf16x4*>(orow + h * DH + f) = pack4(o);
        }
      }
  }
}

// ---- LDS-DMA staging of a streamed 16-row operand tile (flash3_* kernels) -----------------------------------------------------------
// (Round 5, measured and NOT kept: the cached apply and dv sweeps in this form - V / dO tiles double-buffered by DMA, one barrier per
// tile, every wait explicit.  Bit-identical, but at 64 images apply 141 -> 174 us and dv 137 -> 154: their tile body is ~0.5 us, so a
// barrier per tile puts the four waves of a workgroup in lock step with the slowest cache load, where the register-staged 4-tile
// chunks of the flash2 form need a barrier pair per FOUR tiles.  The dk sweep - 3 us per tile, and no registers to prefetch a chunk
// into - is where the DMA staging pays: 296 -> 254 us.)
template <int H, int DH>
struct TileDma {                 // the wave's two 1 KiB pieces of a [16][PITCH] image (see flash3_dk_kernel)
  unsigned v0, v1, lds;          // per-lane source offsets inside a 16-row tile; LDS byte address of the wave's first piece in buffer 0
  __device__ __forceinline__ void init(const unsigned char* buf0, int wave, int lane) {
    typedef FC<H, DH> C;
    constexpr int VPRP = C::PITCH / 8;
    unsigned vv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = (2 * wave + j) * 64 + lane;
      int row = i / VPRP, col = i - row * VPRP;
      if (row > 15) row = 15;
      if (col > C::VPR - 1) col = C::VPR - 1;
      vv[j] = (unsigned)(row * C::D * 2 + col * 16);
    }
    v0 = vv[0]; v1 = vv[1];
    lds = (unsigned)(size_t)buf0 + (unsigned)(2 * wave) * 1024u;
  }
  // rows [16 tile, 16 tile + 16) of the sample's (N x D) matrix at `base` -> buffer `buf` (8 KB apart)
  __device__ __forceinline__ void issue(const char* base, int tile, int buf) const {
    typedef FC<H, DH> C;
    const unsigned long long a = (unsigned long long)(base + (long long)tile * 16 * C::D * 2);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    pc_dma2(reinterpret_cast<const void*>(((unsigned long long)hi << 32) | (unsigned long long)lo), v0, v1,
            (unsigned)__builtin_amdgcn_readfirstlane(lds + (unsigned)buf * 8192u));
  }
};
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// (Round 5, measured and not kept: the cached dq + delta sweep with its K and V tiles staged the same way - double-buffered DMA, one
// barrier per tile, no staging registers, no spills: 317 us against 316 for the flash2 form at 64 images.  That sweep already
// prefetches its one-tile chunks through registers; it is bound by the chain of its tile body, not by staging.)
// dk <- dk - mean over the keys of dk, per (sample, feature).  The softmax is invariant to a common shift of all keys, so
// the exact gradient satisfies sum_k dk[k,:] = 0; the sweeps' dk violates it by the rounding of the dS operand (row sums
// of bf16(dS) are not 0), a component that is pure error - and the one the k convolution's weight gradient, a sum of
// dk x over ALL pixels, picks up coherently when its input has a non-zero mean (un-normalised skip outputs; measured on
// the 512 x 512 configuration's level-1 decoder block: weight-gradient error 0.6 -> 2e-3 of its range).  One workgroup per
// (sample, 64-feature slab): column sums in registers -> LDS -> subtract on a second pass (L2-resident).
// (round 5: 1024 threads = 128 row lanes, the 6 - 7 rows of a thread all in flight, the lanes' sums met by a two-level LDS sum: the
// 256-thread form walked 25 dependent row loads per thread - 18 us at 16 AND at 64 images)
template <int NT>
__global__ __launch_bounds__(NT) void flash_center_dk_kernel(bf16_t* __restrict__ dk, int N, int D) {
  constexpr int RLN = NT / 8, MAXR = 8;
  __shared__ float red[RLN][65];
  __shared__ float red2[8][65];
  const int b = blockIdx.x, slab = blockIdx.y, v = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int f0 = slab * 64 + v * 8;
  bf16_t* base = dk + (long long)b * N * D + f0;
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 0.f;
  const bool live = f0 < D;                       // (D is a multiple of 8; slabs past D do not exist)
  const bool held = N <= MAXR * RLN;              // every row of the thread stays in registers between the two passes
  bf16x8 x[MAXR];
  if (live) {
    if (held) {
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        const int r = rl + k * RLN;
        x[k] = *reinterpret_cast<const bf16x8*>(base + (long long)(r < N ? r : rl) * D);
      }
#pragma unroll
      for (int k = 0; k < MAXR; ++k)
        if (rl + k * RLN < N) {
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] += (float)x[k][i];
        }
    } else {
      for (int r = rl; r < N; r += RLN) {
        const bf16x8 y = *reinterpret_cast<const bf16x8*>(base + (long long)r * D);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += (float)y[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][v * 8 + i] = a[i];
  __syncthreads();
  if (threadIdx.x < 512 && threadIdx.x < 8 * 64) {      // 8 groups of 64 columns: group g sums the row lanes g, g + 8, ...
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    if (g < 8 && (NT >= 512 || g < NT / 64)) {
      float t = 0.f;
      for (int r = g; r < RLN; r += (NT >= 512 ? 8 : NT / 64)) t += red[r][c];
      red2[g][c] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < (NT >= 512 ? 8 : NT / 64); ++g) t += red2[g][threadIdx.x];
    red[0][threadIdx.x] = t / (float)N;
  }
  __syncthreads();
  if (live) {
    if (held) {
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        const int r = rl + k * RLN;
        if (r < N) {
          bf16x8 y = x[k];
#pragma unroll
          for (int i = 0; i < 8; ++i) y[i] = (bf16_t)((float)y[i] - red[0][v * 8 + i]);
          *reinterpret_cast<bf16x8*>(base + (long long)r * D) = y;
        }
      }
    } else {
      for (int r = rl; r < N; r += RLN) {
        bf16x8 y = *reinterpret_cast<const bf16x8*>(base + (long long)r * D);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (bf16_t)((float)y[i] - red[0][v * 8 + i]);
        *reinterpret_cast<bf16x8*>(base + (long long)r * D) = y;
      }
    }
  }
}
int launch_center_dk(const vu_flash_args& a, hipStream_t st) {
  static const bool narrow = [] { const char* e = getenv("VU_CENTER_DK_WIDE"); return e && e[0] == '0'; }();      // A/B switch
  if (narrow) hipLaunchKernelGGL(flash_center_dk_kernel<256>, dim3(a.B, (a.D + 63) / 64), dim3(256), 0, st, (bf16_t*)a.dk, a.N, a.D);
  else hipLaunchKernelGGL(flash_center_dk_kernel<1024>, dim3(a.B, (a.D + 63) / 64), dim3(1024), 0, st, (bf16_t*)a.dk, a.N, a.D);
  if (vu_prof_on()) vu_prof_note("flash_center_dk_kernel", 0.0, 4.0 * (double)a.B * a.N * a.D * 2.0);
  return vu_check_launch("flash_center_dk");
}

// Tail overlap of the backward (round 3).  The dv sweep needs nothing of the dq / dk sweeps and writes only dv, so it can
// be enqueued FIRST, on a LOW-PRIORITY stream forked from `st`, and joined back at the end: the dispatcher serves the dq
// and dk sweeps first and dv's workgroups take the slots their last, partly filled round leaves empty.  Measured on one
// Base level-2 module (N = 784, 13 workgroups per image; forward + backward, us, serial order replayed from a graph ->
// overlapped, eager; tools/flash_bench.py --wall): 40 images 1229 -> 1151, 48: 1286 -> 1253, 56: 1310 -> 1341 (worse),
// 64: 1641 -> 1561, 72: 1672 -> 1678, 80: 1969 -> 1877, 96: 2046 -> 2114 (worse), 128: 2768 -> 2820 (worse); the Base step
// at 64 images 13.42 -> 13.11 ms.  A partly filled round is not a loss by itself - its workgroups have their CU to
// themselves and run ~1.3x faster (two waves per SIMD deliver 1.56x the throughput of one) - so the overlap only pays where
// the serial order starts a nearly EMPTY round: the dq / dk sweeps (two workgroups per CU, 512 slots) with at most 22 % of a
// round left over, or the dv sweep (three per CU, 768 slots) with at most 10 %.  That is the rule below; it is the fit of
// those eight points, nothing deeper.  Never inside a stream capture: a captured kernel node carries no priority
// (hipGraphKernelNodeSetAttribute rejects hipKernelNodeAttributePriority on this runtime), and at EQUAL priority dv's
// workgroups take LDS and slots from the dq sweep from its first round on (64 images: 1942 - 2165 us against 1641).  A caller
// that wants the overlap launches eagerly (vu_model_prefers_eager tells; engine.TrainStep / bench.py follow it).
// VU_FLASH_FORK: unset = that rule; 0 = never; 2 = always when launched eagerly or captured (experiments); 1 = the round-2
// form (dk and dv forked at equal priority after the dq sweep; 2 % slower on the Base step).  Streams and events are
// per-thread library state, created on first use.
struct ForkPool { hipStream_t s[2]; hipEvent_t e[3]; bool ok; int mode; int slots; };
inline ForkPool* fork_pool() {
  static thread_local ForkPool fp = {{nullptr, nullptr}, {nullptr, nullptr, nullptr}, false, 3, 512};
  static thread_local bool tried = false;
  if (!tried) {
    tried = true;
    const char* ev = getenv("VU_FLASH_FORK");
    fp.mode = (ev && ev[0] >= '0' && ev[0] <= '2') ? ev[0] - '0' : 3;
    bool ok = fp.mode != 0;
    int lo = 0, hi = 0, dev = 0, cus = 0;
    if (ok) ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;          // lo: the least priority (largest number)
    if (ok && hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      fp.slots = 2 * cus;
    for (int i = 0; ok && i < 2; ++i)
      ok = (fp.mode == 1 ? hipStreamCreateWithFlags(&fp.s[i], hipStreamNonBlocking) : hipStreamCreateWithPriority(&fp.s[i], hipStreamNonBlocking, lo)) == hipSuccess;
    for (int i = 0; ok && i < 3; ++i) ok = hipEventCreateWithFlags(&fp.e[i], hipEventDisableTiming) == hipSuccess;
    fp.ok = ok;
  }
  return fp.ok ? &fp : nullptr;
}
// does the backward of this shape, launched eagerly on `st`, overlap its dv sweep (see above)?
inline bool tail_overlap(const ForkPool* fp, int nblk, hipStream_t st, bool check_capture) {
  if (!fp || fp->mode == 1) return false;
  if (fp->mode == 2) return true;
  const int r2 = nblk % fp->slots, s3 = fp->slots / 2 * 3, r3 = nblk % s3;
  if (nblk <= fp->slots || !((r2 > 0 && 100 * r2 <= 22 * fp->slots) || (r3 > 0 && 10 * r3 <= s3))) return false;
  if (check_capture) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
  }
  return true;
}

// the same for the 4-head sweeps (3 - 5 workgroups per CU): VU_FLASH_FORK=2 forces it; the default rule is the measurement below
inline bool tail_overlap_v1(const ForkPool* fp, int nblk, hipStream_t st, bool check_capture) {
  if (!fp || fp->mode == 1 || fp->mode == 0) return false;
  static const bool v1_on = [] { const char* e = getenv("VU_FLASH_FORK_V1"); return !(e && e[0] == '0'); }();
  if (fp->mode != 2 && !(v1_on && nblk > 2 * fp->slots)) return false;
  if (check_capture) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
  }
  return true;
}

template <typename K>
int reserve_lds(K kern, size_t lds) {
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    vu_set_error("flash attention: cannot reserve %zu bytes of LDS", lds);
    return VU_ELAUNCH;
  }
  return VU_OK;
}

// profiler tag of a 4-head sweep with its row length ("flash_bwd_dk_kernel<N=3136>"): Lite runs one instantiation at two levels in a
// step, and a mean over both shapes is no launch duration (round-5 review)
static const char* v1_tag(const char* base, int N) {
  static thread_local char buf[96];
  snprintf(buf, sizeof(buf), "%s<N=%d>", base, N);
  return buf;
}

template <int H, int DH>
int launch_backward(const vu_flash_args& a, hipStream_t st) {
  typedef FC<H, DH> C;
#ifndef VU_FLASH_V1_CK2
#define VU_FLASH_V1_CK2 4
#endif
  // CK2: sweeps that keep BOTH stationary tiles in LDS (4 heads of <= 16 features: 2.3 KB per 16-row image, so four streamed tiles
  // per barrier pair still leave four workgroups per CU)
  constexpr int WPB = 4, CK = 4, CK2 = (H == 4 && DH <= 16) ? VU_FLASH_V1_CK2 : 2, NT = H * H + H;      // (dv with 8 tiles per chunk: slower, 2006 vs 2027 images/s)
  const int ntiles = a.N >> 4, per = (ntiles + WPB - 1) / WPB;
  const int nblk = a.B * per;
  const float c = a.scale * 1.44269504088896340736f;
  const size_t rowb = (size_t)16 * C::PITCH * 2;                                      // one 16-row image
  const size_t lds1 = (2 * CK2 + 2 * WPB) * rowb + sizeof(BwdTab<H>) + (size_t)WPB * NT * 4;
  const size_t lds2 = (2 * CK2 + 2 * WPB) * rowb + sizeof(BwdTab<H>) + (size_t)WPB * 2 * H * 16 * 4;
  const size_t lds3 = (2 * CK2 + 2 * WPB) * rowb + (size_t)2 * H * CK2 * 16 * 4 + sizeof(BwdTab<H>) + (size_t)(H * H + H) * 4 + (size_t)WPB * 512;
  const size_t lds4 = (2 * CK) * rowb + (size_t)2 * H * CK * 16 * 4 + sizeof(BwdTab<H>) + (size_t)(H * H + H) * 4 + (size_t)WPB * 512;
  auto k1 = flash_bwd_delta_kernel<H, DH, WPB, CK2>;
  auto k2 = flash_bwd_dq_kernel<H, DH, WPB, CK2>;
  auto k3 = flash_bwd_dkv_kernel<H, DH, WPB, CK2, false>;
  auto k4 = flash_bwd_dkv_kernel<H, DH, WPB, CK, true>;
  VU_TRY(reserve_lds(k1, lds1)); VU_TRY(reserve_lds(k2, lds2)); VU_TRY(reserve_lds(k3, lds3)); VU_TRY(reserve_lds(k4, lds4));
  const double E = (double)a.B * H * a.N * a.N, act = (double)a.B * a.N * C::D * 2.0;
  const bf16_t *q = (const bf16_t*)a.q, *k = (const bf16_t*)a.k, *v = (const bf16_t*)a.v, *dO = (const bf16_t*)a.dO;
  // tail overlap as in the 8-head form (launch_backward_v2): the dv sweep first, on the low-priority stream, joined at the end
  ForkPool* fp = vu_prof_on() ? nullptr : fork_pool();
  hipStream_t s_dv = st;
  const bool early_dv = tail_overlap_v1(fp, nblk, st, true);
  auto launch_dv = [&]() -> int {
    hipLaunchKernelGGL(k4, dim3(nblk), dim3(WPB * 64), lds4, s_dv, q, k, v, dO, a.lse2, (const float*)nullptr, a.stats, (bf16_t*)a.dv, a.B, a.N, c, a.scale, a.rng, 0);
    if (vu_prof_on()) vu_prof_note(v1_tag("flash_bwd_dv_kernel", a.N), 4.0 * E * DH + 2.0 * E * H, 4.0 * act);
    return vu_check_launch("flash_bwd_dv");
  };
  if (early_dv) {
    bool ok = hipEventRecord(fp->e[0], st) == hipSuccess && hipStreamWaitEvent(fp->s[1], fp->e[0], 0) == hipSuccess;
    if (!ok) { vu_set_error("flash attention: stream fork failed"); return VU_ELAUNCH; }
    s_dv = fp->s[1];
    VU_TRY(launch_dv());
  }
  // the training backward behind a forward that saved V = sum_k bf16(P) k: dq leaves the delta sweep (VU_FLASH_FUSE_DQ=0: the
  // separate dq sweep of rounds 2 - 3, for the A/B record)
  static const bool fuse_off = [] { const char* e = getenv("VU_FLASH_FUSE_DQ"); return e && e[0] == '0'; }();
  static const bool clr = [] { const char* e = getenv("VU_FLASH_V1_CLEAR"); return e && e[0] == '1'; }();
  const bool fused = a.pk != nullptr && a.rinv_b != nullptr && a.training && !fuse_off;
  if (fused) {
    auto k1q = flash_bwd_delta_kernel<H, DH, WPB, CK2, true>;
    VU_TRY(reserve_lds(k1q, lds1));
    hipLaunchKernelGGL(k1q, dim3(nblk), dim3(WPB * 64), lds1, st, q, k, v, dO, a.lse2, a.rinv, a.stats, a.delta, a.partials, a.B, a.N, c, a.rng,
                       (const float*)a.pk, (bf16_t*)a.dq, a.scale, (const float*)a.rinv_b, clr ? (int)lds1 : 0);
    if (vu_prof_on()) vu_prof_note(v1_tag("flash_bwd_delta_dq_kernel", a.N), 6.0 * E * DH + 6.0 * E * H, 6.0 * act);
  } else {
    hipLaunchKernelGGL(k1, dim3(nblk), dim3(WPB * 64), lds1, st, q, k, v, dO, a.lse2, a.rinv, a.stats, a.delta, a.partials, a.B, a.N, c, a.rng,
                       (const float*)nullptr, (bf16_t*)nullptr, 0.f, (const float*)nullptr, clr ? (int)lds1 : 0);
    if (vu_prof_on()) vu_prof_note(v1_tag("flash_bwd_delta_kernel", a.N), 4.0 * E * DH + 6.0 * E * H, 4.0 * act);
  }
  VU_TRY(vu_check_launch("flash_bwd_delta"));
  hipLaunchKernelGGL(flash_bwd_mix_finalize_kernel, dim3((unsigned)((H * H + H + 3) / 4)), dim3(256), 0, st, a.partials, nblk, a.stats, a.d_mix_w, a.d_mix_b, H, a.rng.inv_keep);
  VU_TRY(vu_check_launch("flash_bwd_mix_finalize"));
  if (!fused) {
    hipLaunchKernelGGL(k2, dim3(nblk), dim3(WPB * 64), lds2, st, q, k, v, dO, a.lse2, a.delta, a.stats, (bf16_t*)a.dq, a.B, a.N, c, a.scale, a.rng);
    if (vu_prof_on()) vu_prof_note(v1_tag("flash_bwd_dq_kernel", a.N), 6.0 * E * DH + 4.0 * E * H, 5.0 * act);
    VU_TRY(vu_check_launch("flash_bwd_dq"));
  }
  hipLaunchKernelGGL(k3, dim3(nblk), dim3(WPB * 64), lds3, st, q, k, v, dO, a.lse2, a.delta, a.stats, (bf16_t*)a.dk, a.B, a.N, c, a.scale, a.rng, clr ? (int)lds3 : 0);
  if (vu_prof_on()) vu_prof_note(v1_tag("flash_bwd_dk_kernel", a.N), 6.0 * E * DH + 4.0 * E * H, 5.0 * act);
  VU_TRY(vu_check_launch("flash_bwd_dk"));
  VU_TRY(launch_center_dk(a, st));
  if (!early_dv) VU_TRY(launch_dv());
  else if (!(hipEventRecord(fp->e[2], fp->s[1]) == hipSuccess && hipStreamWaitEvent(st, fp->e[2], 0) == hipSuccess)) {
    vu_set_error("flash attention: stream join failed");
    return VU_ELAUNCH;
  }
  {   // bisect switch (tools/contention_ops.py): VU_FLASH_V1_ZERO = bit mask of outputs overwritten with zeros (1 dq, 2 dk, 4 dv)
    static const int zmask = [] { const char* e = getenv("VU_FLASH_V1_ZERO"); return e ? atoi(e) : 0; }();
    const size_t nb = (size_t)a.B * a.N * C::D * 2;
    if (zmask & 1) (void)hipMemsetAsync(a.dq, 0, nb, st);
    if (zmask & 2) (void)hipMemsetAsync(a.dk, 0, nb, st);
    if (zmask & 4) (void)hipMemsetAsync(a.dv, 0, nb, st);
  }
  return VU_OK;
}

template <int H, int DH>
int launch_forward(const vu_flash_args& a, hipStream_t st) {
  typedef FC<H, DH> C;
#ifndef VU_FLASH_V1_CKF
#define VU_FLASH_V1_CKF 8
#endif
  constexpr int WPB = 4, CK = (H == 4 && DH <= 16) ? VU_FLASH_V1_CKF : 4;      // streamed tiles per barrier pair
  const int ntiles = a.N >> 4, per = (ntiles + WPB - 1) / WPB;
  const int nblk = a.B * per;
  const float c = a.scale * 1.44269504088896340736f;
#ifndef VU_FLASH_V1_CKS
#define VU_FLASH_V1_CKS 16
#endif
  constexpr int CKS = (H == 4 && DH <= 16) ? VU_FLASH_V1_CKS : CK;             // (the statistics sweep stages K only)
  const size_t lds1 = (size_t)CKS * 16 * C::PITCH * 2 + (size_t)WPB * C::NMOM * 4;
  const size_t lds2 = (size_t)2 * CK * 16 * C::PITCH * 2 + (size_t)(H * H + H) * 4;
  auto k1 = flash_stats_kernel<H, DH, WPB, CKS>;
  auto k2 = flash_apply_kernel<H, DH, WPB, CK>;
  if (lds1 > 48 * 1024 && hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess) {
    vu_set_error("flash attention: cannot reserve %zu bytes of LDS", lds1); return VU_ELAUNCH;
  }
  if (lds2 > 48 * 1024 && hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) {
    vu_set_error("flash attention: cannot reserve %zu bytes of LDS", lds2); return VU_ELAUNCH;
  }
  const double E = (double)a.B * H * a.N * a.N, act = (double)a.B * a.N * C::D * 2.0;
  hipLaunchKernelGGL(k1, dim3(nblk), dim3(WPB * 64), lds1, st, (const bf16_t*)a.q, (const bf16_t*)a.k, a.lse2, a.rinv, a.partials,
                     (a.training && a.rinv_b) ? a.pk : (float*)nullptr, a.rinv_b, a.B, a.N, c, a.rng, a.training);
  if (vu_prof_on()) vu_prof_note(v1_tag("flash_stats_kernel", a.N), (a.training ? (a.pk ? 3.0 : 2.0) : 1.0) * 2.0 * E * DH, 2.0 * act);
  VU_TRY(vu_check_launch("flash_stats"));
  hipLaunchKernelGGL(flash_bn_finalize_kernel, dim3(1), dim3(1024), 0, st, a.partials, nblk, a.mix_w, a.mix_b, a.bn_w, a.bn_b, a.run_mean,
                     a.run_var, a.stats, H, a.N, (double)a.B * a.N * a.N, a.training, 0.1f, 1e-5f, a.rng.inv_keep, 0);
  VU_TRY(vu_check_launch("flash_bn_finalize"));
  hipLaunchKernelGGL(k2, dim3(nblk), dim3(WPB * 64), lds2, st, (const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, a.lse2,
                     a.stats, (bf16_t*)a.O, a.B, a.N, c, a.rng);
  if (vu_prof_on()) vu_prof_note(v1_tag("flash_apply_kernel", a.N), 4.0 * E * DH + 2.0 * E * H, 4.0 * act);
  return vu_check_launch("flash_apply");
}

template <int DH, int KS, int WPBV = 4>
int launch_forward_v2(const vu_flash_args& a, hipStream_t st) {
#ifndef VU_CKF
#define VU_CKF 4
#endif
#ifndef VU_CKK
#define VU_CKK 2
#endif
  // (eight waves per workgroup - the small-batch form, one workgroup per CU: every barrier pair is fully exposed, so twice the tiles per chunk)
#ifndef VU_CKM
#define VU_CKM 2
#endif
  // streamed tiles per barrier pair.  Round 5: the row-statistics and moments sweeps of the 4-wave form take TWO (the moments sweep
  // sits at 256 registers and a 4-tile chunk costs it 12 more staging registers than a 2-tile one: 214 -> 200 us at 64 images,
  // row statistics 97 -> 91), the apply sweep keeps four (142 against 154 us with two)
  constexpr int H = 8, WPB = WPBV, CK = WPBV == 8 ? 2 * VU_CKF : VU_CKF, CKM = (WPBV == 8 || KS != 1) ? CK : VU_CKM;
  typedef FC<H, DH> C;
  constexpr int TPB = WPB / KS;                                          // own tiles per workgroup (pair_park / pair_take)
  const int ntiles = a.N >> 4, per = (ntiles + TPB - 1) / TPB;
  const int nblk = a.B * per;
  const float c = a.scale * 1.44269504088896340736f;
  const size_t rowb = (size_t)16 * C::PITCH * 2;
  const size_t lds1 = CKM * rowb + (size_t)WPB * C::NMOM * 4;
  size_t ldsm = CKM * rowb + (size_t)WPB * 16 * 4 + (size_t)(7 * DH + 16) * 2 + 16;
  if (KS == 2 && ldsm < (size_t)(WPB / 2) * (H * C::DT + 2) * 1024) ldsm = (size_t)(WPB / 2) * (H * C::DT + 2) * 1024;      // the pair-combine scratch
  // probability cache (round 5): the training forward writes it in the moments sweep and reads it back in the apply sweep
  const bool pc = a.pcache != nullptr && a.training;
  size_t lds2 = (pc ? 1 : 2) * CK * rowb + (size_t)tr_zero_elems<H, DH>() * 2 + (pc ? (size_t)WPB * VU_PC_RING * 4096 : 0);
  if (pc && KS == 2) lds2 = std::max(lds2, (size_t)(WPB / 2) * 4 * (DH / 8) * 1024);      // (pair-combine scratch of the split form)
  auto k1 = flash_rowstats_kernel<H, DH, WPB, CKM, KS>;
  auto km = pc ? flash2_moments_kernel<DH, WPB, CKM, KS, true> : flash2_moments_kernel<DH, WPB, CKM, KS, false>;
  auto k2 = pc ? flash2_apply_kernel<DH, WPB, CK, KS, true> : flash2_apply_kernel<DH, WPB, CK, KS, false>;
  VU_TRY(reserve_lds(k1, lds1)); VU_TRY(reserve_lds(km, ldsm)); VU_TRY(reserve_lds(k2, lds2));
  const double E = (double)a.B * H * a.N * a.N, act = (double)a.B * a.N * C::D * 2.0;
  const bf16_t *q = (const bf16_t*)a.q, *k = (const bf16_t*)a.k, *v = (const bf16_t*)a.v;
  hipLaunchKernelGGL(k1, dim3(nblk), dim3(WPB * 64), lds1, st, q, k, a.lse2, a.rinv, a.B, a.N, c);
  if (vu_prof_on()) vu_prof_note("flash_rowstats_kernel", 4.0 * E * DH, 2.0 * act);
  if (vu_prof_on()) vu_prof_note_strict(2.0 * E * DH);
  VU_TRY(vu_check_launch("flash_rowstats"));
  if (a.training) {
    VU_REQUIRE(a.pk != nullptr, "flash attention: the training forward needs the P k buffer");
    hipLaunchKernelGGL(km, dim3(nblk), dim3(WPB * 64), ldsm, st, q, k, a.lse2, a.mix_w, a.partials, a.pk, a.rinv, a.B, a.N, c, a.rng, a.pcache);
    if (vu_prof_on()) { vu_prof_note("flash2_moments_kernel", 4.0 * E * DH + 2.0 * E * H, 4.0 * act + (pc ? 2.0 * E : 0.0)); vu_prof_note_mapfree(4.0 * act); }
    if (vu_prof_on()) vu_prof_note_strict(2.0 * E * H);
    VU_TRY(vu_check_launch("flash2_moments"));
  }
  hipLaunchKernelGGL(flash_bn_finalize_kernel, dim3(1), dim3(1024), 0, st, a.partials, nblk, a.mix_w, a.mix_b, a.bn_w, a.bn_b, a.run_mean,
                     a.run_var, a.stats, H, a.N, (double)a.B * a.N * a.N, a.training, 0.1f, 1e-5f, a.rng.inv_keep, 1);
  VU_TRY(vu_check_launch("flash_bn_finalize"));
  hipLaunchKernelGGL(k2, dim3(nblk), dim3(WPB * 64), lds2, st, q, k, v, a.lse2, a.stats, (bf16_t*)a.O, a.B, a.N, c, a.rng, (const void*)a.pcache);
  if (vu_prof_on()) { vu_prof_note("flash2_apply_kernel", (pc ? 2.0 : 4.0) * E * DH + 2.0 * E * H, pc ? 2.0 * act + 2.0 * E : 4.0 * act); vu_prof_note_mapfree(pc ? 2.0 * act : 4.0 * act); }
  if (vu_prof_on()) vu_prof_note_strict(2.0 * E * DH);      // (per module 12 E d + 6 E h = 3 x forward; the mix counts once, in the moments sweep)
  return vu_check_launch("flash2_apply");
}

template <int DH, int KS, int WPBV = 4>
int launch_backward_v2(const vu_flash_args& a, hipStream_t st) {
  constexpr int H = 8, WPB = WPBV, CK = 4, CK2 = 2, NT = H * H + H;
  typedef FC<H, DH> C;
  constexpr int TPB = WPB / KS;
  const int ntiles = a.N >> 4, per = (ntiles + TPB - 1) / TPB;
  const int nblk = a.B * per;
  const float c = a.scale * 1.44269504088896340736f;
  const size_t rowb = (size_t)16 * C::PITCH * 2;
  const size_t zr = (size_t)tr_zero_elems<H, DH>() * 2;
  // two workgroups per CU: every one of these must stay <= 80 KB (81920 B)
  const size_t lds1 = (2 * CK2 + 2 * WPB) * rowb + (size_t)WPB * 1280;          // (the final [WPB][NT] reduction aliases the K chunk)
  const size_t lds2 = (2 * CK2 + 2 * WPB) * rowb + zr;
  constexpr int CKK = WPBV == 8 ? 4 : VU_CKK;      // (eight waves, one workgroup per CU: the whole 160 KB, four tiles per barrier pair)
  size_t lds3 = (2 * CKK + 2 * WPB) * rowb + (size_t)WPB * 1280;
  static_assert((size_t)WPB * NT * 4 <= (size_t)CK2 * 16 * C::PITCH * 2, "reduction scratch must fit the K chunk");
  const size_t lds4 = (2 * CK + 2 * WPB) * rowb + (size_t)2 * H * CK * 16 * 4 + (size_t)WPB * 1024;
  constexpr int CKX = WPBV == 8 ? 4 : (KS == 2 ? 2 : 1);       // (split form: a chunk must hold a tile for each wave of a pair)
  const size_t lds2xr = (2 * CKX + 2 * WPB) * rowb + (size_t)WPB * 1024 + (size_t)tr_strip_elems<DH>() * 2;
  auto k1 = flash2_bwd_delta_kernel<DH, WPB, CK2>;
  auto k2 = flash2_bwd_dq_kernel<DH, WPB, CK2>;
  const bool pc = a.pcache != nullptr && a.training && a.pk != nullptr;      // (the cache holds what the TRAINING forward's moments sweep packed)
  size_t lds2x = pc ? (2 * CKX + WPB) * rowb + (size_t)WPB * 1024 + (size_t)tr_strip_elems<DH>() * 2 : lds2xr;
  if (pc && KS == 2) lds2x = std::max(lds2x, (size_t)(WPB / 2) * (4 * (DH / 8) + 4) * 1024);
  auto k2x = pc ? flash2_bwd_dqx_kernel<DH, WPB, CKX, KS, true> : flash2_bwd_dqx_kernel<DH, WPB, CKX, KS, false>;
  // The fused sweep takes V = sum_k P k from the TRAINING forward (flash2_moments_kernel); with running statistics (eval
  // mode + autograd) no moments sweep ran and pk holds nothing, so that case takes the separate delta and dq sweeps.
  // (VU_FLASH_UNFUSED=1, read once: diagnostic switch that takes the separate sweeps in training too)
  static const bool unfused_dbg = [] { const char* e = getenv("VU_FLASH_UNFUSED"); return e && e[0] == '1'; }();
  const bool fused = a.pk != nullptr && a.training && !unfused_dbg;
  if (KS != 1 && !fused) { vu_set_error("flash attention: the split backward exists for the fused training form only"); return VU_EUNSUPPORTED; }
  VU_TRY(reserve_lds(k2x, lds2x));
#ifndef VU_PC_CKV
#define VU_PC_CKV 4
#endif
  constexpr int CKV = WPBV == 8 ? 4 : (KS == 2 ? 2 : 1);   // dv: one tile per chunk, 16 stationary rows per wave: 43.5 KB, three workgroups per CU (split form: two)
  // cached form: the dv sweep stages dO only and keeps no stationary rows - four tiles per chunk (register-prefetched) still leave
  // three workgroups per CU
  constexpr int CKVP = WPBV == 8 ? 4 : (KS == 2 ? 2 : VU_PC_CKV);
  auto k3 = pc ? flash2_bwd_dkv_kernel<DH, WPB, CKK, false, KS, true> : flash2_bwd_dkv_kernel<DH, WPB, CKK, false, KS, false>;
  auto k4 = pc ? flash2_bwd_dkv_kernel<DH, WPB, CKVP, true, KS, true> : flash2_bwd_dkv_kernel<DH, WPB, CKV, true, KS, false>;
  size_t lds3v = pc ? CKVP * rowb + (size_t)WPB * 5120 + (size_t)WPB * VU_PC_RING * 4096 : (2 * CKV + WPB) * rowb + (size_t)WPB * 1280;
  if (pc) lds3 = (2 * CKK + WPB) * rowb + (size_t)WPB * 5120;
  if (pc && KS == 2) {      // the pair-combine scratch at the end of a split sweep (pair_park / pair_take) aliases the chunk area: it must fit
    const size_t pair = (size_t)(WPB / 2) * H * C::DT * 1024;
    lds3v = std::max(lds3v, pair); lds3 = std::max(lds3, pair);
  }
  VU_TRY(reserve_lds(k1, lds1)); VU_TRY(reserve_lds(k2, lds2)); VU_TRY(reserve_lds(k3, lds3)); VU_TRY(reserve_lds(k4, lds3v));
  (void)lds4;
  const double E = (double)a.B * H * a.N * a.N, act = (double)a.B * a.N * C::D * 2.0;
  const bf16_t *q = (const bf16_t*)a.q, *k = (const bf16_t*)a.k, *v = (const bf16_t*)a.v, *dO = (const bf16_t*)a.dO;
  ForkPool* fp = vu_prof_on() ? nullptr : fork_pool();
  hipStream_t s_dk = st, s_dv = st;
  const bool early_dv = tail_overlap(fp, nblk, st, true);
  auto launch_dv = [&]() -> int {
    hipLaunchKernelGGL(k4, dim3(nblk), dim3(WPB * 64), lds3v, s_dv, q, k, v, dO, a.lse2, (const float*)nullptr, a.stats, (bf16_t*)a.dv, a.B, a.N, c, a.scale, a.rng,
                       (const void*)a.pcache);
    if (vu_prof_on()) { vu_prof_note("flash2_bwd_dv_kernel", (pc ? 2.0 : 4.0) * E * DH + 2.0 * E * H, pc ? 2.0 * act + 2.0 * E : 4.0 * act); vu_prof_note_mapfree(pc ? 2.0 * act : 4.0 * act); }
    if (vu_prof_on()) vu_prof_note_strict(2.0 * E * DH);
    return vu_check_launch("flash2_bwd_dv");
  };
  if (early_dv) {
    bool ok = hipEventRecord(fp->e[0], st) == hipSuccess && hipStreamWaitEvent(fp->s[1], fp->e[0], 0) == hipSuccess;
    if (!ok) { vu_set_error("flash attention: stream fork failed"); return VU_ELAUNCH; }
    s_dv = fp->s[1];
    VU_TRY(launch_dv());
  }
  if (fused) {
    hipLaunchKernelGGL(k2x, dim3(nblk), dim3(WPB * 64), lds2x, st, q, k, v, dO, a.lse2, a.rinv, a.pk, a.stats, (bf16_t*)a.dq, a.delta, a.partials, a.B,
                       a.N, c, a.scale, a.rng, a.training ? 0 : 1, (const void*)a.pcache);
    if (vu_prof_on()) { vu_prof_note("flash2_bwd_dqx_kernel", (pc ? 4.0 : 6.0) * E * DH + 8.0 * E * H, pc ? 6.0 * act + 2.0 * E : 7.0 * act); vu_prof_note_mapfree(pc ? 6.0 * act : 7.0 * act); }
    if (vu_prof_on()) vu_prof_note_strict(4.0 * E * DH + 4.0 * E * H);      // dA^ = dO v^T, dq = dS k, dP~ = W^T e, dW = e P~^T
    VU_TRY(vu_check_launch("flash2_bwd_dqx"));
  } else {
    hipLaunchKernelGGL(k1, dim3(nblk), dim3(WPB * 64), lds1, st, q, k, v, dO, a.lse2, a.rinv, a.stats, a.delta, a.partials, a.B, a.N, c, a.rng);
    if (vu_prof_on()) vu_prof_note("flash2_bwd_delta_kernel", 4.0 * E * DH + 6.0 * E * H, 4.0 * act);
    VU_TRY(vu_check_launch("flash2_bwd_delta"));
  }
  hipLaunchKernelGGL(flash_bwd_mix_finalize_kernel, dim3((unsigned)((H * H + H + 3) / 4)), dim3(256), 0, st, a.partials, nblk, a.stats, a.d_mix_w, a.d_mix_b, H, a.rng.inv_keep);
  VU_TRY(vu_check_launch("flash_bwd_mix_finalize"));
  const bool late_fork = fp && fp->mode == 1;
  if (late_fork) {
    bool ok = hipEventRecord(fp->e[0], st) == hipSuccess;
    ok = ok && hipStreamWaitEvent(fp->s[0], fp->e[0], 0) == hipSuccess && hipStreamWaitEvent(fp->s[1], fp->e[0], 0) == hipSuccess;
    if (!ok) { vu_set_error("flash attention: stream fork failed"); return VU_ELAUNCH; }
    s_dk = fp->s[0]; s_dv = fp->s[1];
  }
  if (!fused) {
    hipLaunchKernelGGL(k2, dim3(nblk), dim3(WPB * 64), lds2, st, q, k, v, dO, a.lse2, a.delta, a.stats, (bf16_t*)a.dq, a.B, a.N, c, a.scale, a.rng);
    if (vu_prof_on()) vu_prof_note("flash2_bwd_dq_kernel", 6.0 * E * DH + 4.0 * E * H, 5.0 * act);
    VU_TRY(vu_check_launch("flash2_bwd_dq"));
  }
  // (VU_FLASH_DK3=0, read once: the flash2 kernel with synchronous chunk staging, for the A/B record)
  static const bool dk3_off = [] { const char* e = getenv("VU_FLASH_DK3"); return e && e[0] == '0'; }();
  if (pc && KS == 1 && WPB == 4 && DH == 24 && !dk3_off) {
    auto k3d = flash3_dk_kernel<24>;
    constexpr size_t lds3d = 4 * 8192 + (size_t)WPB * 16 * FC<8, 24>::PITCH * 2 + (size_t)WPB * 5120;
    VU_TRY(reserve_lds(k3d, lds3d));
    hipLaunchKernelGGL(k3d, dim3(nblk), dim3(256), lds3d, s_dk, q, v, dO, a.delta, a.stats, (bf16_t*)a.dk, a.B, a.N, a.scale, (const void*)a.pcache);
  } else
  hipLaunchKernelGGL(k3, dim3(nblk), dim3(WPB * 64), lds3, s_dk, q, k, v, dO, a.lse2, a.delta, a.stats, (bf16_t*)a.dk, a.B, a.N, c, a.scale, a.rng, (const void*)a.pcache);
  if (vu_prof_on()) vu_prof_note((pc && KS == 1 && WPB == 4 && DH == 24 && !dk3_off) ? "flash3_dk_kernel" : "flash2_bwd_dk_kernel",
                                 (pc ? 4.0 : 6.0) * E * DH + 4.0 * E * H, pc ? 4.0 * act + 2.0 * E : 5.0 * act);
  if (vu_prof_on()) { vu_prof_note_mapfree(pc ? 4.0 * act : 5.0 * act); vu_prof_note_strict(2.0 * E * DH); }
  VU_TRY(vu_check_launch("flash2_bwd_dk"));
  VU_TRY(launch_center_dk(a, s_dk));
  if (!early_dv) VU_TRY(launch_dv());
  if (early_dv || late_fork) {
    bool ok = (!late_fork || hipEventRecord(fp->e[1], fp->s[0]) == hipSuccess) && hipEventRecord(fp->e[2], fp->s[1]) == hipSuccess;
    ok = ok && (!late_fork || hipStreamWaitEvent(st, fp->e[1], 0) == hipSuccess) && hipStreamWaitEvent(st, fp->e[2], 0) == hipSuccess;
    if (!ok) { vu_set_error("flash attention: stream join failed"); return VU_ELAUNCH; }
  }
  return VU_OK;
}

}  // namespace

#define VU_FLASH_DISPATCH(FN, ...)                                                     \
  do {                                                                                 \
    const int dh_ = a.D / a.H;                                                         \
    if (a.H == 4 && dh_ == 32) return FN<4, 32>(__VA_ARGS__);                         \
    if (a.H == 4 && dh_ == 16) return FN<4, 16>(__VA_ARGS__);                         \
    if (a.H == 4 && dh_ == 48) return FN<4, 48>(__VA_ARGS__);                         \
    vu_set_error("flash attention: shape H=%d d=%d not instantiated", a.H, dh_);      \
    return VU_EUNSUPPORTED;                                                            \
  } while (0)

bool vu_flash_tail_overlap(int B, int N, int H) {
  if (H != 8) return tail_overlap_v1(fork_pool(), B * (((N >> 4) + 3) / 4), nullptr, false);
  return tail_overlap(fork_pool(), B * (((N >> 4) + 3) / 4), nullptr, false);
}

bool vu_flash_pays(int B, int N) { return (long long)B * ((N / 16 + 3) / 4) >= 192; }

bool vu_flash_ok(int dtype, int B, int N, int D, int H) {
  if (dtype != 1 || H <= 0 || D % H != 0) return false;
  const int dh = D / H;
  const bool inst = (H == 8 && (dh == 24 || dh == 8 || dh == 32)) || (H == 4 && (dh == 32 || dh == 16 || dh == 48));
  // the dropout word index of a map element (one word per 4 keys) must fit 32 bits: B H N N < 2^34
  return inst && N % 16 == 0 && N >= 256 && (double)B * H * N * N < 17179869184.0;
}

// Split of the streamed axis over wave pairs ("split of the streamed axis" above).  MEASURED, NOT TAKEN BY DEFAULT (round 3):
// one Base level-2 module, forward + backward, unsplit -> split: 16 images 735 -> 735 us (dq+delta 147 -> 145, dk 131 -> 119,
// moments 77 -> 67, dv 83 -> 98..111, apply / row statistics unchanged), 8 images 694 -> 578.  The two waves of a pair run in
// lock-step through the two barriers of every streamed chunk and both still do their share of the chunk staging, so the second
// wave on the SIMD hides little (two INDEPENDENT workgroups per CU deliver 1.56x); and where it does gain (8 - 12 images) the
// materialising kernels are still faster (Base step at 8 images: 5.82 ms materialised, 6.02 split, 6.49 unsplit; at 12:
// 6.32 / 6.52 / 6.62).  So the rule is "never"; vu_set_flash_key_split(2) / VU_FLASH_KS=2 run it (tests keep it correct).
static int g_key_split = [] { const char* e = getenv("VU_FLASH_KS"); return (e && e[0] >= '1' && e[0] <= '3') ? e[0] - '0' : 0; }();
extern "C" int vu_set_flash_key_split(int ks) {
  if (ks < 0 || ks > 3) { vu_set_error("vu_set_flash_key_split: 0 (default), 1, 2 or 3 (2 with eight waves per workgroup)"); return VU_EINVAL; }
  g_key_split = ks;
  return VU_OK;
}
// Default (no setter, no environment): the unsplit sweeps, except where they would put a single wave on every SIMD - at most
// one 4-wave workgroup per CU (B ceil(N / 64) <= 256: Base / Large level 2 at 16 images per GPU, the per-GPU batch of BASELINE
// configs 3 - 4).  There the split form with EIGHT waves per workgroup runs (form 3: the same 13 workgroups per sample, each
// tile's keys shared by a wave pair, two waves per SIMD): one level-2 module, forward + backward, 16 images: 730 -> 677 us
// (tools/flash_bench.py; form 2 - the split with four waves per workgroup - 730; at 32 images form 3 is slower: 1024 -> 1145).
int vu_flash_key_split(int B, int N) {
  if (g_key_split) return g_key_split;
  return (long long)B * ((N / 16 + 3) / 4) <= 256 ? 3 : 1;
}

// Probability cache of the 8-head form: process-level switch (like the attention form: it changes the workspace layout, so set it
// between steps, before the workspace is sized).  VU_FLASH_PCACHE=0 / 1, read once; default in VU_FLASH_PCACHE_DEFAULT.
#ifndef VU_FLASH_PCACHE_DEFAULT
#define VU_FLASH_PCACHE_DEFAULT 1
#endif
static int g_pcache = [] { const char* e = getenv("VU_FLASH_PCACHE"); return e ? (e[0] == '0' ? 0 : 1) : VU_FLASH_PCACHE_DEFAULT; }();
extern "C" int vu_set_flash_pcache(int on) {
  if (on < -1 || on > 1) { vu_set_error("vu_set_flash_pcache: -1 (default), 0 or 1"); return VU_EINVAL; }
  g_pcache = on < 0 ? VU_FLASH_PCACHE_DEFAULT : on;
  return VU_OK;
}
// Budget of ONE model workspace for its probability caches (sum over the attention modules, bytes): modules are granted a cache in
// carve order while the sum fits; the rest run the recompute sweeps.  Default 96 GiB (a third of the 288 GB of an MI355X: Base at
// 64 images takes 2.5 GiB, 512 x 512 at 32 images 37 GB); VU_FLASH_PCACHE_BUDGET_MB, read once; vu_set_flash_pcache_budget().
static size_t g_pcache_budget = [] {
  const char* e = getenv("VU_FLASH_PCACHE_BUDGET_MB");
  return e ? (size_t)strtoull(e, nullptr, 10) << 20 : (size_t)96 << 30;
}();
extern "C" int vu_set_flash_pcache_budget(unsigned long long bytes) { g_pcache_budget = (size_t)bytes; return VU_OK; }
size_t vu_flash_pcache_budget() { return g_pcache_budget; }
size_t vu_flash_pcache_bytes(int B, int N, int D, int H) {
  if (!g_pcache || H != 8 || !vu_flash_ok(1, B, N, D, H)) return 0;
  const size_t nt = (size_t)(N >> 4);
  return (size_t)B * nt * nt * 4096;
}

size_t vu_flash_partials_floats(int B, int N, int H) {
  const int ntiles = N >> 4, per = (ntiles + 1) / 2;          // (rows of the split form: two own tiles per workgroup)
  return (size_t)B * per * (H * H + H) + 64;          // forward: H + H (H + 1) / 2 moments; backward: H H + H mix-gradient sums
}

int vu_k_flash_forward(const vu_flash_args& a, hipStream_t st) {
  VU_REQUIRE(vu_flash_ok(1, a.B, a.N, a.D, a.H), "flash attention: shape not covered");
  if (a.H == 8) {       // head mixes on the matrix cores (v2 tile body)
    const int dh = a.D / a.H;
    if (vu_flash_key_split(a.B, a.N) == 3 && dh == 24) return launch_forward_v2<24, 2, 8>(a, st);        // (Base / Large level 2 only)
    if (vu_flash_key_split(a.B, a.N) >= 2) {
      if (dh == 24) return launch_forward_v2<24, 2>(a, st);
      if (dh == 8) return launch_forward_v2<8, 2>(a, st);
      if (dh == 32) return launch_forward_v2<32, 2>(a, st);
    }
    if (dh == 24) return launch_forward_v2<24, 1>(a, st);
    if (dh == 8) return launch_forward_v2<8, 1>(a, st);
    if (dh == 32) return launch_forward_v2<32, 1>(a, st);
  }
  VU_FLASH_DISPATCH(launch_forward, a, st);
}

int vu_k_flash_backward(const vu_flash_args& a, hipStream_t st) {
  VU_REQUIRE(vu_flash_ok(1, a.B, a.N, a.D, a.H), "flash attention: shape not covered");
  if (a.H == 8) {
    const int dh = a.D / a.H;
    if (vu_flash_key_split(a.B, a.N) == 3 && dh == 24 && a.pk != nullptr && a.training) return launch_backward_v2<24, 2, 8>(a, st);
    if (vu_flash_key_split(a.B, a.N) >= 2 && a.pk != nullptr && a.training) {      // (the eval-mode backward keeps the unsplit sweeps)
      if (dh == 24) return launch_backward_v2<24, 2>(a, st);
      if (dh == 8) return launch_backward_v2<8, 2>(a, st);
      if (dh == 32) return launch_backward_v2<32, 2>(a, st);
    }
    if (dh == 24) return launch_backward_v2<24, 1>(a, st);
    if (dh == 8) return launch_backward_v2<8, 1>(a, st);
    if (dh == 32) return launch_backward_v2<32, 1>(a, st);
  }
  VU_FLASH_DISPATCH(launch_backward, a, st);
}

vu_rng vu_flash_quad_rng(vu_rng r) {
  if (r.thr) {
    const uint32_t t8 = (r.thr + 128u) >> 8;         // thr is round(65536 p)
    r.thr = t8 ? (t8 > 255u ? 255u : t8) : 1u;       // (p >= 0.998 would give 256: every byte dropped, 1 / keep = inf)
    r.inv_keep = 256.0f / (256.0f - (float)r.thr);
  }
  return r;
}
