// Common device/host helpers for the MI355X (gfx950) ViT-UNet kernels.
// Wave = 64 lanes everywhere.  Storage type T is float (fp32 parity mode) or __bf16 (bf16
// training mode); all arithmetic is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define VU_OK 0
#define VU_EINVAL (-1)        // bad argument / shape
#define VU_EUNSUPPORTED (-2)  // configuration outside what the kernels cover
#define VU_EWORKSPACE (-3)    // workspace too small
#define VU_ELAUNCH (-4)       // hip launch error (see vu_last_error)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

void vu_set_error(const char* fmt, ...);
int vu_check_launch(const char* what);
// profiler annotation for the NEXT vu_check_launch: kernel tag + algorithmic flops / bytes
void vu_prof_note(const char* tag, double flops, double bytes);
void vu_prof_note_mapfree(double bytes_mapfree);     // after vu_prof_note: bytes without map-sized streams (the probability cache)
void vu_prof_note_strict(double flops_strict);      // after vu_prof_note: flops without recomputation / padding
bool vu_prof_on();

#define VU_REQUIRE(cond, ...)                  \
  do {                                         \
    if (!(cond)) {                             \
      vu_set_error(__VA_ARGS__);               \
      return VU_EINVAL;                        \
    }                                          \
  } while (0)

static inline int vu_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline long long vu_cdiv64(long long a, long long b) { return (a + b - 1) / b; }
static inline size_t vu_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// storage <-> float
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float vu_ld(const float* p) { return *p; }
__device__ __forceinline__ float vu_ld(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void vu_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void vu_st(bf16_t* p, float v) { *p = (bf16_t)v; }

// 4-element vector access (all activation rows / map rows are 4-element aligned by layout).
struct vu_f4 { float v[4]; };
__device__ __forceinline__ vu_f4 vu_ld4(const float* p) {
  float4 t = *reinterpret_cast<const float4*>(p);
  return {{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ vu_f4 vu_ld4(const bf16_t* p) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  vu_f4 r;
  r.v[0] = __uint_as_float(t.x << 16);
  r.v[1] = __uint_as_float(t.x & 0xffff0000u);
  r.v[2] = __uint_as_float(t.y << 16);
  r.v[3] = __uint_as_float(t.y & 0xffff0000u);
  return r;
}
// 8 consecutive elements (16-byte aligned for bf16): one 16-byte load instead of two 8-byte ones
__device__ __forceinline__ void vu_ld8(const float* p, vu_f4& lo, vu_f4& hi) { lo = vu_ld4(p); hi = vu_ld4(p + 4); }
__device__ __forceinline__ void vu_ld8(const bf16_t* p, vu_f4& lo, vu_f4& hi) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  lo.v[0] = __uint_as_float(t.x << 16); lo.v[1] = __uint_as_float(t.x & 0xffff0000u);
  lo.v[2] = __uint_as_float(t.y << 16); lo.v[3] = __uint_as_float(t.y & 0xffff0000u);
  hi.v[0] = __uint_as_float(t.z << 16); hi.v[1] = __uint_as_float(t.z & 0xffff0000u);
  hi.v[2] = __uint_as_float(t.w << 16); hi.v[3] = __uint_as_float(t.w & 0xffff0000u);
}
__device__ __forceinline__ void vu_st4(float* p, const vu_f4& a) {
  *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
__device__ __forceinline__ void vu_st4(bf16_t* p, const vu_f4& a) {
  bf16x4 t = {(bf16_t)a.v[0], (bf16_t)a.v[1], (bf16_t)a.v[2], (bf16_t)a.v[3]};
  *reinterpret_cast<bf16x4*>(p) = t;
}

// ---------------------------------------------------------------------------------------------
// counter-based dropout RNG (replayed bit for bit by oracle/vit_unet_oracle.py: keep_mask).
// One 32-bit hash serves two neighbouring elements (16 bits each).
// ---------------------------------------------------------------------------------------------
struct vu_rng {
  uint32_t k0, k1;
  uint32_t thr;  // drop when r16 < thr ; thr = round(p * 65536)
  float inv_keep;
  const uint32_t* salt;  // optional device word mixed into the key at kernel start (lets a
                         // captured hipGraph draw fresh masks on every replay); null in tests
};
__device__ __forceinline__ vu_rng vu_rng_resolve(vu_rng r) {
  if (r.salt) { const uint32_t s = *r.salt; r.k0 ^= s * 0x9e3779b9u; r.k1 += s; }
  return r;
}
__host__ __device__ __forceinline__ uint32_t vu_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t vu_hash_pair(const vu_rng& r, uint64_t idx) {
  uint64_t w = idx >> 1;
  return vu_mix32((uint32_t)w ^ r.k0) + (uint32_t)(w >> 32) * 0x9e3779b9u + r.k1;
}
__host__ __device__ __forceinline__ bool vu_keep(const vu_rng& r, uint64_t idx) {
  uint32_t x = vu_hash_pair(r, idx);
  uint32_t v = (idx & 1) ? (x >> 16) : (x & 0xffffu);
  return v >= r.thr;
}
// 32-bit fast path for an even element index < 2^32: one hash word serves elements idx, idx+1
__device__ __forceinline__ uint32_t vu_hash_word32(const vu_rng& r, uint32_t idx_even) {
  return vu_mix32((idx_even >> 1) ^ r.k0) + r.k1;
}
static inline uint64_t vu_splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline vu_rng vu_make_rng(uint64_t seed, uint64_t stream, float p) {
  uint64_t k = vu_splitmix64(seed ^ vu_splitmix64(stream));
  vu_rng r;
  r.k0 = (uint32_t)k;
  r.k1 = (uint32_t)(k >> 32);
  r.thr = p > 0.f ? (uint32_t)(p * 65536.0f + 0.5f) : 0u;
  r.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  r.salt = nullptr;
  return r;
}

// ---------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float vu_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float vu_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block sum for blockDim.x <= 1024; `sm` must hold >= 16 floats; result valid in all threads.
__device__ __forceinline__ float vu_block_sum(float v, float* sm) {
  v = vu_wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sm[i];
  return t;
}

__device__ __forceinline__ float vu_gelu(float x) {  // exact erf GELU (torch.nn.GELU default)
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float vu_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
