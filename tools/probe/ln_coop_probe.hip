// Probe: the block LayerNorm forward (K13: statistics over all P elements of an image, elementwise affine) as ONE kernel with a
// per-image barrier among the 8 workgroups of an image, against the library's two launches (chunk statistics, apply).  Measures both
// with HIP events over back-to-back launches and checks the outputs bit for bit.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/ln_coop_probe.hip -o tools/probe/ln_coop_probe && ./tools/probe/ln_coop_probe [B = 64] [P = 150528]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16_t;
constexpr int CH = 4096, G = 8, MAXC = 5;

__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  union U8 { uint4 u; bf16_t h[8]; } t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t.h[e] = (bf16_t)v[e];
  return t.u;
}
__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
// chunk partial (count, mean, M2) of the 2 x 256 x 8 values a workgroup holds
__device__ __forceinline__ void chunk_partial(const uint4 (&t)[2], long long base, long long P, float* sm, float* o) {
  float v[2][8];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    unpack8(t[it], v[it]);
    if (e < P) {
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += v[it][q];
    }
  }
  const float tot = block_sum(sum, sm);
  long long n = P - base; if (n > CH) n = CH;
  const float mean = tot / (float)n;
  float m2 = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    if (e < P) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float d = v[it][q] - mean; m2 += d * d; }
    }
  }
  const float M2 = block_sum(m2, sm);
  if (threadIdx.x == 0) { o[0] = (float)n; o[1] = mean; o[2] = M2; }
}
template <bool COHERENT>
__device__ __forceinline__ float ldp(const float* p) {
  if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COHERENT>
__device__ __forceinline__ void merge_stats(const float* partials, int b, int nch, float eps, float* sm2, float& mean, float& rstd) {
  if (threadIdx.x < 64) {
    double n = 0.0, s = 0.0;
    for (int c = threadIdx.x; c < nch; c += 64) {
      const float* o = partials + ((long long)b * nch + c) * 3;
      const float o0 = ldp<COHERENT>(o), o1 = ldp<COHERENT>(o + 1);
      n += (double)o0; s += (double)o0 * (double)o1;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { n += __shfl_xor(n, m, 64); s += __shfl_xor(s, m, 64); }
    const double mu = s / n;
    double M2 = 0.0;
    for (int c = threadIdx.x; c < nch; c += 64) {
      const float* o = partials + ((long long)b * nch + c) * 3;
      const double d = (double)ldp<COHERENT>(o + 1) - mu;
      M2 += (double)ldp<COHERENT>(o + 2) + (double)ldp<COHERENT>(o) * d * d;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) M2 += __shfl_xor(M2, m, 64);
    if (threadIdx.x == 0) { sm2[0] = (float)mu; sm2[1] = (float)(1.0 / sqrt(M2 / n + (double)eps)); }
  }
  __syncthreads();
  mean = sm2[0]; rstd = sm2[1];
}
__device__ __forceinline__ uint4 apply8(const uint4& tz, const float* w, const float* bias, long long e, float mean, float rstd) {
  float t[8];
  unpack8(tz, t);
  const float4 w0 = *reinterpret_cast<const float4*>(w + e), w1 = *reinterpret_cast<const float4*>(w + e + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(bias + e), b1 = *reinterpret_cast<const float4*>(bias + e + 4);
  const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int q = 0; q < 8; ++q) t[q] = (t[q] - mean) * rstd * wv[q] + bv[q];
  return pack8(t);
}

// ---- the two launches of the library -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_kernel(const bf16_t* a, float* partials, long long P) {
  __shared__ float sm[16];
  const int c = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
  const long long base = (long long)c * CH, sb = (long long)b * P;
  uint4 t[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    t[it] = e < P ? *reinterpret_cast<const uint4*>(a + sb + e) : make_uint4(0, 0, 0, 0);
  }
  chunk_partial(t, base, P, sm, partials + ((long long)b * nch + c) * 3);
}
__global__ __launch_bounds__(256) void apply_kernel(const bf16_t* __restrict__ z, const float* __restrict__ w, const float* __restrict__ bias,
                                                    bf16_t* __restrict__ y, const float* partials, float* stats, long long P, float eps) {
  __shared__ float sm2[2];
  const int c = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
  const long long base = (long long)c * CH, sb = (long long)b * P;
  uint4 tz[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    tz[it] = e < P ? *reinterpret_cast<const uint4*>(z + sb + e) : make_uint4(0, 0, 0, 0);
  }
  float mean, rstd;
  merge_stats<false>(partials, b, nch, eps, sm2, mean, rstd);
  if (c == 0 && threadIdx.x == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    if (e < P) *reinterpret_cast<uint4*>(y + sb + e) = apply8(tz[it], w, bias, e, mean, rstd);
  }
}

// ---- one launch: workgroup (image b, member g) holds chunks g, g + 8, ... in registers across a barrier of the image's 8 members ------
// blockIdx -> (b, g): workgroups id and id + 8 share an XCD (round-robin dispatch), so the 8 members of an image are placed on ONE
// XCD (one L2); the fences below are agent scope all the same, correctness does not rest on the placement.
template <int MODE>      // 0: agent-scope fences; 1: atomics only (measurement: right only while the members share an L2); 2: no barrier (wrong, measurement)
__global__ __launch_bounds__(256) void coop_kernel(const bf16_t* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                                                   bf16_t* __restrict__ y, float* partials, float* stats, unsigned long long* cnt, int B,
                                                   long long P, int nch, float eps, int* err) {
  __shared__ float sm[16];
  __shared__ float sm2[2];
  const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3, g = j & 7, b = (j >> 3) * 8 + xcd;
  if (b >= B) return;
  const long long sb = (long long)b * P;
  uint4 t[MAXC][2];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const long long base = (long long)(g + G * k) * CH;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const long long e = base + (it * 256 + threadIdx.x) * 8;
      t[k][it] = e < P ? *reinterpret_cast<const uint4*>(a + sb + e) : make_uint4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = g + G * k;
    if (c < nch) chunk_partial(t[k], (long long)c * CH, P, sm, partials + ((long long)b * nch + c) * 3);
  }
  if (MODE < 2 && threadIdx.x == 0) {
    if (MODE == 0) __threadfence();                                        // release: this member's partials
    else __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long old = atomicAdd(cnt + b, 1ull);
    const unsigned long long target = old - old % G + G;                   // launches on one counter are stream-ordered: it starts at a multiple of G
    int spins = 0;
    while (__hip_atomic_load(cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 24)) { *err = 1; break; }                        // (never in a healthy run: a member that was not dispatched)
    }
    if (MODE == 0) __threadfence();                                        // acquire
  }
  __syncthreads();
  float mean, rstd;
  merge_stats<true>(partials, b, nch, eps, sm2, mean, rstd);
  if (g == 0 && threadIdx.x == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const long long base = (long long)(g + G * k) * CH;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const long long e = base + (it * 256 + threadIdx.x) * 8;
      if (e < P) *reinterpret_cast<uint4*>(y + sb + e) = apply8(t[k][it], w, bias, e, mean, rstd);
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  const long long P = argc > 2 ? atoll(argv[2]) : 150528;
  const int nch = (int)((P + CH - 1) / CH), reps = 200;
  if (nch > G * MAXC || nch < G) { printf("P out of the probe's range\n"); return 1; }
  std::vector<unsigned short> ha((size_t)B * P);
  std::vector<float> hw(P), hb(P);
  unsigned s = 12345;
  for (auto& v : ha) { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 9) % 2001 - 1000) / 500.0f; v = (unsigned short)(__builtin_bit_cast(unsigned, f) >> 16); }
  for (long long i = 0; i < P; ++i) { hw[i] = 0.5f + (i % 7) * 0.1f; hb[i] = (i % 5) * 0.05f - 0.1f; }
  bf16_t *a, *y0, *y1; float *w, *bias, *part0, *part1, *st0, *st1; unsigned long long* cnt; int* err;
  CK(hipMalloc(&a, (size_t)B * P * 2)); CK(hipMalloc(&y0, (size_t)B * P * 2)); CK(hipMalloc(&y1, (size_t)B * P * 2));
  CK(hipMalloc(&w, P * 4)); CK(hipMalloc(&bias, P * 4)); CK(hipMalloc(&part0, (size_t)B * nch * 12)); CK(hipMalloc(&part1, (size_t)B * nch * 12));
  CK(hipMalloc(&st0, B * 8)); CK(hipMalloc(&st1, B * 8)); CK(hipMalloc(&cnt, 8 * 1024)); CK(hipMalloc(&err, 4));
  CK(hipMemset(cnt, 0, 8 * 1024)); CK(hipMemset(err, 0, 4));
  CK(hipMemcpy(a, ha.data(), (size_t)B * P * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), P * 4, hipMemcpyHostToDevice));
  const int ngrp = (B + 7) / 8, grid_coop = ngrp * 64;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int pass = 0; pass < 2; ++pass) {
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(stats_kernel, dim3(nch, B), dim3(256), 0, 0, a, part0, P);
      hipLaunchKernelGGL(apply_kernel, dim3(nch, B), dim3(256), 0, 0, a, w, bias, y0, part0, st0, P, 1e-5f);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) printf("B=%d P=%lld two launches : %.2f us per LayerNorm\n", B, P, 1e3 * ms / reps);
    for (int mode = 2; mode >= 0; --mode) {
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) {
        if (mode == 0) hipLaunchKernelGGL(coop_kernel<0>, dim3(grid_coop), dim3(256), 0, 0, a, w, bias, y1, part1, st1, cnt, B, P, nch, 1e-5f, err);
        if (mode == 1) hipLaunchKernelGGL(coop_kernel<1>, dim3(grid_coop), dim3(256), 0, 0, a, w, bias, y1, part1, st1, cnt, B, P, nch, 1e-5f, err);
        if (mode == 2) hipLaunchKernelGGL(coop_kernel<2>, dim3(grid_coop), dim3(256), 0, 0, a, w, bias, y1, part1, st1, cnt, B, P, nch, 1e-5f, err);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      if (pass) printf("B=%d P=%lld one launch, mode %d : %.2f us per LayerNorm\n", B, P, mode, 1e3 * ms / reps);
    }
  }
  std::vector<unsigned short> h0((size_t)B * P), h1((size_t)B * P);
  int herr = 0;
  CK(hipMemcpy(h0.data(), y0, (size_t)B * P * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, (size_t)B * P * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  long long diff = 0;
  for (size_t i = 0; i < h0.size(); ++i) diff += h0[i] != h1[i];
  printf("outputs differ in %lld of %zu elements; barrier timeouts: %d\n", diff, h0.size(), herr);
  return diff != 0 || herr;
}
