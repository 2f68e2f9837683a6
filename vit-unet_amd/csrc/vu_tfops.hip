// Element-wise / row-wise ops of the reference's Keras re-implementation (the "TF-only" layers):
//   /root/reference/vit_unet/tf/functions.py:60-132   Resampling: 'max' / 'avg' token pooling
//   /root/reference/vit_unet/tf/functions.py:258-311  AttentionTransformerEncoder: Keras MultiHeadAttention (softmax over
//                                                     keys, dropout on the probabilities), LayerNormalization over the last
//                                                     axis with epsilon 1e-3, FeedForward with a GELU after BOTH Dense layers
//   /root/reference/vit_unet/tf/model.py:208          input residual Y = X + unpatch(...)
// None of them is on the benchmarked path: plain streaming kernels (16-byte accesses where the layout allows, one wave per
// softmax row), all arithmetic fp32, storage fp32 or bf16.  TensorFlow is not in this image: parity is against the oracle's
// restatement of the reference TEXT (unpinned).
#include "vu_kernels.h"
#include "../../include/vit_unet_amd.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n4) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    vu_f4 v = vu_ld4(x + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) v.v[j] = vu_gelu(v.v[j]);
    vu_st4(y + 4 * i, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long long n4) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const vu_f4 v = vu_ld4(x + 4 * i), g = vu_ld4(dy + 4 * i);
    vu_f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = g.v[j] * vu_gelu_grad(v.v[j]);
    vu_st4(dx + 4 * i, o);
  }
}

// Resampling 'max' / 'avg' (tf/functions.py:101-124).  The reference pools pairs of consecutive tokens (MaxPool1D /
// AveragePooling1D, pool 2, stride 2 over the token axis), reshapes (B, N/2, P) to (B, N/4, 2, P), pools pairs again along
// the N/4 axis inside a map over the batch, and concatenates the two slices of the middle axis back along the tokens.
// Followed through index by index (pool_size = 4, the ratio of every adjacent pair of patch sizes in tf/model.py:12):
//   out[b, w + s N/8, :] = pool over the source tokens 8 w + 2 s + {0, 1, 4, 5}        (s in {0,1}, w < N/8)
// then + position_embedding[m, :].  (Not a 2 x 2 spatial pool of neighbouring patches - the text pools along the row-major
// token order - and that is what is restated.)
__device__ __forceinline__ int pool4_src(int m, int N8) { const int s = m / N8, w = m - s * N8; return 8 * w + 2 * s; }

template <typename T, bool MAX>
__global__ __launch_bounds__(256) void pool4_fwd_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ y,
                                                        int N, int P4, long long total) {
  const int N8 = N >> 3, M = N >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p4 = (int)(i % P4);
    const long long bm = i / P4;
    const int m = (int)(bm % M);
    const long long b = bm / M;
    const T* src = x + ((b * N + pool4_src(m, N8)) * (long long)P4 + p4) * 4;
    const long long st = (long long)P4 * 4;
    const vu_f4 a0 = vu_ld4(src), a1 = vu_ld4(src + st), a2 = vu_ld4(src + 4 * st), a3 = vu_ld4(src + 5 * st);
    vu_f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o.v[j] = MAX ? fmaxf(fmaxf(a0.v[j], a1.v[j]), fmaxf(a2.v[j], a3.v[j])) : 0.25f * ((a0.v[j] + a1.v[j]) + (a2.v[j] + a3.v[j]));
      if (pos) o.v[j] += pos[((long long)m * P4 + p4) * 4 + j];
    }
    vu_st4(y + i * 4, o);
  }
}
// dx of the pooling: 'avg' spreads dy / 4; 'max' routes dy to the first maximum in the order 0, 1, 4, 5 (TensorFlow's
// max-pool gradient picks one maximal element; ties do not occur on continuous data)
template <typename T, bool MAX>
__global__ __launch_bounds__(256) void pool4_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                        int N, int P4, long long total) {
  const int N8 = N >> 3, M = N >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p4 = (int)(i % P4);
    const long long bm = i / P4;
    const int m = (int)(bm % M);
    const long long b = bm / M;
    const long long base = ((b * N + pool4_src(m, N8)) * (long long)P4 + p4) * 4;
    const long long st = (long long)P4 * 4;
    const vu_f4 g = vu_ld4(dy + i * 4);
    vu_f4 o[4];
    if (MAX) {
      const vu_f4 a0 = vu_ld4(x + base), a1 = vu_ld4(x + base + st), a2 = vu_ld4(x + base + 4 * st), a3 = vu_ld4(x + base + 5 * st);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float mx = fmaxf(fmaxf(a0.v[j], a1.v[j]), fmaxf(a2.v[j], a3.v[j]));
        const int k = a0.v[j] == mx ? 0 : (a1.v[j] == mx ? 1 : (a2.v[j] == mx ? 2 : 3));
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q].v[j] = q == k ? g.v[j] : 0.f;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[q].v[j] = 0.25f * g.v[j];
    }
    vu_st4(dx + base, o[0]); vu_st4(dx + base + st, o[1]); vu_st4(dx + base + 4 * st, o[2]); vu_st4(dx + base + 5 * st, o[3]);
  }
}

// Keras MultiHeadAttention core (tf/functions.py:288-293 -> keras MultiHeadAttention._compute_attention): P = softmax(scale s)
// over the keys, Pd = dropout(P).  One wave per row; the mask index of element (row, j) is row * ld + j (vu_keep, replayed by
// the oracle).  Both P (the backward needs the un-dropped probabilities) and Pd (the operand of the value product) are written.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const T* __restrict__ s, T* __restrict__ p, T* __restrict__ pd, long long rows,
                                                               int n, int ld, float scale, vu_rng rng_in) {
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int lane = threadIdx.x & 63;
  for (long long row = blockIdx.x * 4LL + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const T* sr = s + row * ld;
    float mx = -3.0e38f;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, vu_ld(sr + j) * scale);
    mx = vu_wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += __expf(vu_ld(sr + j) * scale - mx);
    sum = vu_wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < n; j += 64) {
      const float pr = __expf(vu_ld(sr + j) * scale - mx) * inv;
      vu_st(p + row * ld + j, pr);
      const bool keep = rng.thr == 0 || vu_keep(rng, (uint64_t)row * ld + j);
      vu_st(pd + row * ld + j, keep ? pr * rng.inv_keep : 0.f);
    }
  }
}
// ds = scale * P (g - sum_k P g), g = mask / keep * dPd
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dpd, T* __restrict__ ds, long long rows,
                                                               int n, int ld, float scale, vu_rng rng_in) {
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int lane = threadIdx.x & 63;
  for (long long row = blockIdx.x * 4LL + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    float dot = 0.f;
    for (int j = lane; j < n; j += 64) {
      const bool keep = rng.thr == 0 || vu_keep(rng, (uint64_t)row * ld + j);
      if (keep) dot += vu_ld(p + row * ld + j) * vu_ld(dpd + row * ld + j) * rng.inv_keep;
    }
    dot = vu_wave_sum(dot);
    for (int j = lane; j < n; j += 64) {
      const bool keep = rng.thr == 0 || vu_keep(rng, (uint64_t)row * ld + j);
      const float g = keep ? vu_ld(dpd + row * ld + j) * rng.inv_keep : 0.f;
      vu_st(ds + row * ld + j, scale * vu_ld(p + row * ld + j) * (g - dot));
    }
  }
}

inline unsigned grid_cap(long long work, int per_block) {
  long long g = (work + per_block - 1) / per_block;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

#define VU_TF_DTYPE() VU_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)")

extern "C" {

int vu_add(int dtype, const void* a, const void* b, void* out, long long n, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE(n >= 0 && n % 4 == 0, "vu_add: n must be a multiple of 4");
  return vu_k_add(dtype, a, b, out, n, (hipStream_t)stream);
}

int vu_dropout(int dtype, const void* in, void* out, long long n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE(n >= 0 && n % 4 == 0 && p >= 0.f && p < 1.f, "vu_dropout: n must be a multiple of 4, p in [0, 1)");
  return vu_k_dropout(dtype, in, out, n, vu_make_rng(seed, stream_id, p), (hipStream_t)stream);
}

int vu_gelu_fwd(int dtype, const void* x, void* y, long long n, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE(n >= 0 && n % 4 == 0, "vu_gelu_fwd: n must be a multiple of 4");
  if (n == 0) return VU_OK;
  if (dtype == 0) hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(grid_cap(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n / 4);
  else hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(grid_cap(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n / 4);
  return vu_check_launch("vu_gelu_fwd");
}
int vu_gelu_bwd(int dtype, const void* x, const void* dy, void* dx, long long n, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE(n >= 0 && n % 4 == 0, "vu_gelu_bwd: n must be a multiple of 4");
  if (n == 0) return VU_OK;
  if (dtype == 0) hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(grid_cap(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, (float*)dx, n / 4);
  else hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(grid_cap(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n / 4);
  return vu_check_launch("vu_gelu_bwd");
}

int vu_token_pool4_fwd(int dtype, int mode, const void* x, const float* pos, void* y, int B, int N, int P, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE((mode == 0 || mode == 1) && B > 0 && N > 0 && N % 8 == 0 && P > 0 && P % 4 == 0, "vu_token_pool4: mode 0 (max) / 1 (avg), N % 8 == 0, P % 4 == 0");
  const long long total = (long long)B * (N / 4) * (P / 4);
  const dim3 g(grid_cap(total, 256)), t(256);
  hipStream_t st = (hipStream_t)stream;
#define VU_POOL_F(T, MAX) hipLaunchKernelGGL((pool4_fwd_kernel<T, MAX>), g, t, 0, st, (const T*)x, pos, (T*)y, N, P / 4, total)
  if (dtype == 0) { if (mode == 0) VU_POOL_F(float, true); else VU_POOL_F(float, false); }
  else { if (mode == 0) VU_POOL_F(bf16_t, true); else VU_POOL_F(bf16_t, false); }
#undef VU_POOL_F
  return vu_check_launch("vu_token_pool4_fwd");
}
int vu_token_pool4_bwd(int dtype, int mode, const void* x, const void* dy, void* dx, int B, int N, int P, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE((mode == 0 || mode == 1) && B > 0 && N > 0 && N % 8 == 0 && P > 0 && P % 4 == 0, "vu_token_pool4: mode 0 (max) / 1 (avg), N % 8 == 0, P % 4 == 0");
  const long long total = (long long)B * (N / 4) * (P / 4);
  const dim3 g(grid_cap(total, 256)), t(256);
  hipStream_t st = (hipStream_t)stream;
#define VU_POOL_B(T, MAX) hipLaunchKernelGGL((pool4_bwd_kernel<T, MAX>), g, t, 0, st, (const T*)x, (const T*)dy, (T*)dx, N, P / 4, total)
  if (dtype == 0) { if (mode == 0) VU_POOL_B(float, true); else VU_POOL_B(float, false); }
  else { if (mode == 0) VU_POOL_B(bf16_t, true); else VU_POOL_B(bf16_t, false); }
#undef VU_POOL_B
  return vu_check_launch("vu_token_pool4_bwd");
}

int vu_softmax_rows_fwd(int dtype, const void* s, void* p, void* pd, long long rows, int n, int ld, float scale, float p_drop,
                        uint64_t seed, uint64_t stream_id, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE(rows > 0 && n > 0 && ld >= n && p_drop >= 0.f && p_drop < 1.f, "vu_softmax_rows: bad shape or dropout");
  const vu_rng rng = vu_make_rng(seed, stream_id, p_drop);
  const dim3 g(grid_cap(rows, 4)), t(256);
  if (dtype == 0) hipLaunchKernelGGL(softmax_rows_fwd_kernel<float>, g, t, 0, (hipStream_t)stream, (const float*)s, (float*)p, (float*)pd, rows, n, ld, scale, rng);
  else hipLaunchKernelGGL(softmax_rows_fwd_kernel<bf16_t>, g, t, 0, (hipStream_t)stream, (const bf16_t*)s, (bf16_t*)p, (bf16_t*)pd, rows, n, ld, scale, rng);
  return vu_check_launch("vu_softmax_rows_fwd");
}
int vu_softmax_rows_bwd(int dtype, const void* p, const void* dpd, void* ds, long long rows, int n, int ld, float scale, float p_drop,
                        uint64_t seed, uint64_t stream_id, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE(rows > 0 && n > 0 && ld >= n && p_drop >= 0.f && p_drop < 1.f, "vu_softmax_rows: bad shape or dropout");
  const vu_rng rng = vu_make_rng(seed, stream_id, p_drop);
  const dim3 g(grid_cap(rows, 4)), t(256);
  if (dtype == 0) hipLaunchKernelGGL(softmax_rows_bwd_kernel<float>, g, t, 0, (hipStream_t)stream, (const float*)p, (const float*)dpd, (float*)ds, rows, n, ld, scale, rng);
  else hipLaunchKernelGGL(softmax_rows_bwd_kernel<bf16_t>, g, t, 0, (hipStream_t)stream, (const bf16_t*)p, (const bf16_t*)dpd, (bf16_t*)ds, rows, n, ld, scale, rng);
  return vu_check_launch("vu_softmax_rows_bwd");
}

int vu_add_layernorm_fwd_eps(int dtype, const void* a, const void* x, void* z, const float* w, const float* b, void* y,
                             float* ws, float* stats, int B, long long P, float eps, void* stream) {
  VU_TF_DTYPE();
  VU_REQUIRE(eps > 0.f, "vu_add_layernorm_fwd_eps: eps must be positive");
  return vu_k_add_ln_fwd(dtype, a, x, z, w, b, y, ws, stats, B, P, eps, (hipStream_t)stream);
}

}  // extern "C"
