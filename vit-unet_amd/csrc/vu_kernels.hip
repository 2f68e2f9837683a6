// HBM-bound kernels of the ViT-UNet path for gfx950: re-tiling, per-patch 3x3 convolutions,
// row softmax + dropout, head mixing + BatchNorm on attention maps, (N,D) LayerNorm, loss,
// AdamW.  Storage T in {float, __bf16}; arithmetic fp32.  Reference call sites are cited per
// kernel (file:line into /root/reference/vit_unet/torch/model.py unless noted).
#include "vu_kernels.h"

#define VU_DISPATCH_T(dtype, ...)                  \
  if ((dtype) == 0) { typedef float T; __VA_ARGS__ } \
  else { typedef bf16_t T; __VA_ARGS__ }

static inline int grid_for(long long work_items, int block = 256, int cap = 256 * 16) {
  long long g = (work_items + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// =============================================================================================
// a1-a4 retile (model.py:8-53).  SURVEY App. A index maps.  4 consecutive x per thread.
// =============================================================================================
template <typename TI, typename TO>
__global__ void retile_kernel(const TI* __restrict__ in, TO* __restrict__ out,
                              const float* __restrict__ pos, long long total4, int P, int C, int im,
                              int s_in, int s_out) {
  const int e_in = im / s_in, e_out = im / s_out;
  const int ss_in = s_in * s_in, ss_out = s_out * s_out;
  const int D_in = C * ss_in, D_out = C * ss_out;
  const int P4 = P >> 2;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total4;
       t += (long long)gridDim.x * blockDim.x) {
    const unsigned t32 = (unsigned)t;                 // launcher guarantees total4 < 2^32
    const long long b = t32 / (unsigned)P4;
    const int r = (int)(t32 - (unsigned)b * (unsigned)P4) << 2;
    const int n_out = r / D_out, f = r - n_out * D_out;
    const int ch = f / ss_out, rem = f - ch * ss_out;
    const int i = rem / s_out, j = rem - i * s_out;
    const int y = (n_out / e_out) * s_out + i, x = (n_out % e_out) * s_out + j;
    const int n_in = (y / s_in) * e_in + x / s_in;
    const int f_in = ch * ss_in + (y % s_in) * s_in + (x % s_in);
    vu_f4 v = vu_ld4(in + b * P + (long long)n_in * D_in + f_in);
    if (pos) {
      const float4 pp = *reinterpret_cast<const float4*>(pos + r);
      v.v[0] += pp.x; v.v[1] += pp.y; v.v[2] += pp.z; v.v[3] += pp.w;
    }
    vu_st4(out + b * P + r, v);
  }
}

// pure permutation of 16-byte pieces (bf16 -> bf16, no positional add, both patch sizes multiples of 8): half the
// instructions and index divisions per byte of the 4-element form
// ADD: out = retile(in) + add (add in the OUTPUT tiling; the sum in fp32, rounded once - what vu_k_add would leave): the gradient
// of a down-sampling meets the gradient that arrives through the skip connection in the same pass (round 6: two launches less).
template <bool ADD>
__global__ __launch_bounds__(256) void retile_copy16_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long long total8, int P,
                                                             int C, int im, int s_in, int s_out, const uint4* __restrict__ add) {
  const int e_in = im / s_in, e_out = im / s_out;
  const int ss_in = s_in * s_in, ss_out = s_out * s_out;
  const int D_in = C * ss_in, D_out = C * ss_out;
  const int P8 = P >> 3;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total8; t += (long long)gridDim.x * blockDim.x) {
    const unsigned t32 = (unsigned)t;
    const unsigned b = t32 / (unsigned)P8;
    const int r = (int)(t32 - b * (unsigned)P8) << 3;
    const int n_out = r / D_out, f = r - n_out * D_out;
    const int ch = f / ss_out, rem = f - ch * ss_out;
    const int i = rem / s_out, j = rem - i * s_out;
    const int y = (n_out / e_out) * s_out + i, x = (n_out % e_out) * s_out + j;
    const int n_in = (y / s_in) * e_in + x / s_in;
    const int f_in = ch * ss_in + (y % s_in) * s_in + (x % s_in);
    uint4 v = in[((long long)b * P + (long long)n_in * D_in + f_in) >> 3];
    if constexpr (ADD) {
      const uint4 a = add[((long long)b * P + r) >> 3];
      const unsigned vw[4] = {v.x, v.y, v.z, v.w}, aw[4] = {a.x, a.y, a.z, a.w};
      union { uint4 u; bf16_t h[8]; } o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o.h[2 * e] = (bf16_t)(__uint_as_float(vw[e] << 16) + __uint_as_float(aw[e] << 16));
        o.h[2 * e + 1] = (bf16_t)(__uint_as_float(vw[e] & 0xffff0000u) + __uint_as_float(aw[e] & 0xffff0000u));
      }
      v = o.u;
    }
    out[((long long)b * P + r) >> 3] = v;
  }
}

int vu_k_retile(int dtype, int in_f32, int out_f32, const void* in, void* out, const float* pos,
                int B, int C, int im, int s_in, int s_out, hipStream_t st, const void* add) {
  VU_REQUIRE(s_in % 4 == 0 && s_out % 4 == 0 && im % s_in == 0 && im % s_out == 0,
             "vu_retile: patch sizes must be multiples of 4 that divide im (im=%d s_in=%d s_out=%d)", im, s_in, s_out);
  const int P = C * im * im;
  const long long total4 = (long long)B * (P / 4);
  if (total4 == 0) return VU_OK;
  VU_REQUIRE(total4 < 4294967295LL, "vu_retile: more than 2^32 element quads");
  const int grid = grid_for(total4);
  const bool fi = in_f32 || dtype == 0, fo = out_f32 || dtype == 0;
  if (!fi && !fo && !pos && s_in % 8 == 0 && s_out % 8 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)add & 15) == 0) {
    if (add) hipLaunchKernelGGL(retile_copy16_kernel<true>, dim3(grid_for(total4 / 2)), dim3(256), 0, st, (const uint4*)in, (uint4*)out, total4 / 2, P, C, im, s_in, s_out, (const uint4*)add);
    else hipLaunchKernelGGL(retile_copy16_kernel<false>, dim3(grid_for(total4 / 2)), dim3(256), 0, st, (const uint4*)in, (uint4*)out, total4 / 2, P, C, im, s_in, s_out, (const uint4*)nullptr);
    if (vu_prof_on()) vu_prof_note("retile_kernel", 0.0, (double)total4 * 4 * (add ? 3 : 2) * 2.0);
    return vu_check_launch("vu_retile");
  }
  if (add) {      // (no fused form for this layout: the permutation, then the sum in place)
    if (int e = vu_k_retile(dtype, in_f32, out_f32, in, out, pos, B, C, im, s_in, s_out, st, nullptr)) return e;
    VU_REQUIRE(!fo || dtype == 0, "vu_retile: an addend needs the output in the storage type");
    return vu_k_add(dtype, out, add, out, (long long)B * P, st);
  }
  if (fi && fo) hipLaunchKernelGGL((retile_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)in, (float*)out, pos, total4, P, C, im, s_in, s_out);
  else if (fi) hipLaunchKernelGGL((retile_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)in, (bf16_t*)out, pos, total4, P, C, im, s_in, s_out);
  else if (fo) hipLaunchKernelGGL((retile_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)in, (float*)out, pos, total4, P, C, im, s_in, s_out);
  else hipLaunchKernelGGL((retile_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, pos, total4, P, C, im, s_in, s_out);
  if (vu_prof_on()) vu_prof_note("retile_kernel", 0.0, (double)total4 * 4 * 2 * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_retile");
}

struct HeadPadArgs { const bf16_t* src[4]; bf16_t* dst[4]; };
__global__ __launch_bounds__(256) void head_pad_kernel(const HeadPadArgs a, long long groups, int gs, int gd) {
  const bf16_t* __restrict__ src = a.src[blockIdx.y];
  bf16_t* __restrict__ dst = a.dst[blockIdx.y];
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < groups; t += (long long)gridDim.x * blockDim.x) {
    const long long rh = t / gd;                       // (row, head)
    const int g = (int)(t - rh * gd);                  // 4-feature group of the destination head
    uint2 v = make_uint2(0u, 0u);
    if (g < gs) v = *reinterpret_cast<const uint2*>(src + (rh * gs + g) * 4);
    *reinterpret_cast<uint2*>(dst + t * 4) = v;
  }
}
int vu_k_head_pad(const void* const* src, void* const* dst, int n, long long rows, int H, int dh_src, int dh_dst, hipStream_t st) {
  VU_REQUIRE(n >= 1 && n <= 4 && rows > 0 && H > 0 && dh_src > 0 && dh_dst > 0 && dh_src % 4 == 0 && dh_dst % 4 == 0, "vu_head_pad: bad argument");
  HeadPadArgs a;
  for (int i = 0; i < 4; ++i) { a.src[i] = (const bf16_t*)(i < n ? src[i] : src[0]); a.dst[i] = (bf16_t*)(i < n ? dst[i] : dst[0]); }
  for (int i = 0; i < n; ++i) VU_REQUIRE(a.src[i] && a.dst[i] && !(((uintptr_t)a.src[i] | (uintptr_t)a.dst[i]) & 7), "vu_head_pad: null or misaligned tensor");
  const long long groups = rows * H * (dh_dst / 4);
  hipLaunchKernelGGL(head_pad_kernel, dim3(grid_for(groups), n), dim3(256), 0, st, a, groups, dh_src / 4, dh_dst / 4);
  if (vu_prof_on()) vu_prof_note("head_pad_kernel", 0.0, (double)n * rows * H * (dh_src + dh_dst) * 2.0);
  return vu_check_launch("vu_head_pad");
}

template <typename T>
__global__ void batch_sum_kernel(const T* __restrict__ in, float* __restrict__ out, int B, long long P) {
  const long long r = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4;
  if (r >= P) return;
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int b = 0; b < B; ++b) {
    vu_f4 v = vu_ld4(in + b * P + r);
    a0 += v.v[0]; a1 += v.v[1]; a2 += v.v[2]; a3 += v.v[3];
  }
  float4* o = reinterpret_cast<float4*>(out + r);
  float4 c = *o;
  c.x += a0; c.y += a1; c.z += a2; c.w += a3;
  *o = c;
}
int vu_k_batch_sum(int dtype, const void* in, float* out, int B, long long P, hipStream_t st) {
  VU_REQUIRE(P % 4 == 0, "vu_batch_sum: P %% 4 != 0");
  const int grid = vu_cdiv(P / 4, 256);
  VU_DISPATCH_T(dtype, hipLaunchKernelGGL((batch_sum_kernel<T>), dim3(grid), dim3(256), 0, st, (const T*)in, out, B, P);)
  return vu_check_launch("vu_batch_sum");
}

// =============================================================================================
// K7+K8 softmax + dropout (model.py:156-157).  One wave per row; probabilities are stored
// sign-tagged: +p kept, -p dropped (p > 0 always), so every later pass recovers both the
// pre-dropout probability |p| and the mask without re-running the RNG.
// =============================================================================================
template <typename T>
__global__ void softmax_dropout_kernel(T* S, long long rows, int N, int ld, vu_rng rng_in) {
  const vu_rng rng = vu_rng_resolve(rng_in);
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  T* p = S + row * ld;
  float mx = -INFINITY;
  for (int j = lane; j < N; j += 64) mx = fmaxf(mx, vu_ld(p + j));
  mx = vu_wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < N; j += 64) sum += __expf(vu_ld(p + j) - mx);
  sum = vu_wave_sum(sum);
  const float inv = 1.0f / sum;
  const uint64_t base = (uint64_t)row * (uint64_t)ld;   // mask index of (row, j) = row * ld + j
  for (int j = lane; j < N; j += 64) {
    float v = __expf(vu_ld(p + j) - mx) * inv;
    if (rng.thr && !vu_keep(rng, base + j)) v = -v;
    vu_st(p + j, v);
  }
  for (int j = N + lane; j < ld; j += 64) vu_st(p + j, 0.f);
}
int vu_k_softmax_dropout(int dtype, void* S, long long rows, int N, int ld, vu_rng rng, hipStream_t st) {
  if (rows == 0) return VU_OK;
  const long long grid = (rows + 3) / 4;
  VU_REQUIRE(grid < 2147483647LL, "vu_softmax_dropout: too many rows");
  VU_DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_dropout_kernel<T>), dim3((unsigned)grid), dim3(256), 0, st, (T*)S, rows, N, ld, rng);)
  if (vu_prof_on()) vu_prof_note("softmax_dropout_kernel", 0.0, (double)rows * N * 2 * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_softmax_dropout");
}

// =============================================================================================
// K9+K10 head mixing (1x1 conv across heads, model.py:135,159) + BatchNorm2d(h) (:136,159).
// Train: pass 1 accumulates per-channel shifted moments of A_g = sum_h W[g,h] P~_h + c_g, the
// finalize kernel folds mean / rstd / gamma / beta into Wf, cf so that pass 2 is a plain mix.
// =============================================================================================
template <typename T, int H>
__device__ __forceinline__ void load_heads4(const T* base, long long head_stride, float inv_keep,
                                            float (&pt)[H][4], float (&pa)[H][4]) {
  // pt = post-dropout probability (0 if dropped, |p|/keep otherwise) ; pa = |p|
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const vu_f4 v = vu_ld4(base + h * head_stride);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pa[h][e] = fabsf(v.v[e]);
      pt[h][e] = v.v[e] > 0.f ? v.v[e] * inv_keep : 0.f;
    }
  }
}

template <typename T, int H>
__global__ __launch_bounds__(256) void mix_stats_kernel(const T* __restrict__ Ps, const float* __restrict__ W,
                                                        const float* __restrict__ c, float* partials,
                                                        int B, int N, int ld, float inv_keep) {
  // W is wave-uniform and indexed with compile-time constants: scalar loads, no LDS / VGPR copy
  __shared__ float red[4][2 * H];
  float shift[H];   // shift_g = sum_h W[g,h] / N  (the exact mean without dropout; the bias cancels)
#pragma unroll
  for (int g = 0; g < H; ++g) {
    float sacc = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) sacc += W[g * H + h];
    shift[g] = sacc / (float)N;
  }
  const int ld4 = ld >> 2;
  const long long total = (long long)B * N * ld4;
  const long long hs = (long long)N * ld;
  float s1[H], s2[H];
#pragma unroll
  for (int g = 0; g < H; ++g) { s1[g] = 0.f; s2[g] = 0.f; }
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    // 32-bit index arithmetic (launcher guarantees < 2^32 vector positions): 64-bit divisions by
    // run-time values cost more than the mix itself
    const unsigned t32 = (unsigned)t;
    const unsigned row = t32 / (unsigned)ld4;
    const int jc = (int)(t32 - row * (unsigned)ld4) << 2;
    const long long b = row / (unsigned)N;
    const int i = (int)(row - (unsigned)b * (unsigned)N);
    float pt[H][4], pa[H][4];
    load_heads4<T, H>(Ps + (b * H * N + i) * (long long)ld + jc, hs, inv_keep, pt, pa);
#pragma unroll
    for (int g = 0; g < H; ++g) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = -shift[g];
#pragma unroll
        for (int h = 0; h < H; ++h) a += W[g * H + h] * pt[h][e];
        if (jc + e < N) { s1[g] += a; s2[g] += a * a; }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < H; ++g) {
    const float a = vu_wave_sum(s1[g]), b2 = vu_wave_sum(s2[g]);
    if (lane == 0) { red[wave][g] = a; red[wave][H + g] = b2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * H)
    partials[blockIdx.x * 2 * H + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// single block of 256 threads; H <= 16.  stats layout: Wf[H*H] cf[H] mean[H] rstd[H] m1[H] m2[H]
// All 2H partial columns are reduced side by side: thread t sums column t % 2H over blocks
// t / 2H, t / 2H + 256 / 2H, ... (coalesced), then an LDS tree per column, in fp64.
__global__ void bn_finalize_kernel(const float* partials, int nblocks, const float* W, const float* c,
                                   const float* gamma, const float* beta, float* run_mean, float* run_var,
                                   float* stats, int H, int N, double count, int training, float momentum, float eps) {
  __shared__ double sd[256];
  __shared__ float smom[32];
  const int Q = 2 * H, per = 256 / Q;          // Q <= 32 columns, `per` row lanes
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
  double a = 0.0;
  if (training && rl < per) {
    // independent loads first (8 in flight), then the fp64 sums: this single block is pure latency
    int i = rl;
    for (; i + 7 * per < nblocks; i += 8 * per) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partials[(i + u * per) * Q + q];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += (double)t[u];
    }
    for (; i < nblocks; i += per) a += (double)partials[i * Q + q];
  }
  sd[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < Q) {
    double t = 0.0;
    for (int r = 0; r < per; ++r) t += sd[r * Q + threadIdx.x];
    smom[threadIdx.x] = (float)(t / count);
  }
  __syncthreads();
  if (threadIdx.x < H) {
    const int g = threadIdx.x;
    float mean, var;
    if (training) {
      float sw = 0.f;
      for (int h = 0; h < H; ++h) sw += W[g * H + h];
      const float shift = c[g] + sw / (float)N;
      const float d1 = smom[g], d2 = smom[H + g];
      mean = shift + d1;
      var = fmaxf(d2 - d1 * d1, 0.f);
      const float unb = count > 1.0 ? (float)(count / (count - 1.0)) : 1.f;
      run_mean[g] = (1.f - momentum) * run_mean[g] + momentum * mean;
      run_var[g] = (1.f - momentum) * run_var[g] + momentum * var * unb;
    } else {
      mean = run_mean[g];
      var = run_var[g];
    }
    const float rstd = rsqrtf(var + eps);
    const float sc = gamma[g] * rstd;
    for (int h = 0; h < H; ++h) stats[g * H + h] = W[g * H + h] * sc;
    stats[H * H + g] = (c[g] - mean) * sc + beta[g];
    stats[H * H + H + g] = mean;
    stats[H * H + 2 * H + g] = rstd;
    stats[H * H + 3 * H + g] = 0.f;
    stats[H * H + 4 * H + g] = 0.f;
    {  // centred-map form: Ahat_g = sc_g (a_g - shift0_g) + kappa_g with shift0 = sum_h W[g,h] / N
      float sw = 0.f;
      for (int h = 0; h < H; ++h) sw += W[g * H + h];
      stats[VU_BN_STATS_SC(H) + g] = sc;
      stats[VU_BN_STATS_SC(H) + H + g] = (c[g] - mean) * sc + beta[g] + sc * (sw / (float)N);
    }
  }
}

template <typename T, int H>
__global__ __launch_bounds__(256) void mix_apply_kernel(const T* __restrict__ Ps, T* __restrict__ Ah,
                                                        const float* __restrict__ stats, int B, int N, int ld,
                                                        float inv_keep) {
  const int ld4 = ld >> 2;
  const long long total = (long long)B * N * ld4;
  const long long hs = (long long)N * ld;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    // 32-bit index arithmetic (launcher guarantees < 2^32 vector positions): 64-bit divisions by
    // run-time values cost more than the mix itself
    const unsigned t32 = (unsigned)t;
    const unsigned row = t32 / (unsigned)ld4;
    const int jc = (int)(t32 - row * (unsigned)ld4) << 2;
    const long long b = row / (unsigned)N;
    const int i = (int)(row - (unsigned)b * (unsigned)N);
    const long long off = (b * H * N + i) * (long long)ld + jc;
    float pt[H][4], pa[H][4];
    load_heads4<T, H>(Ps + off, hs, inv_keep, pt, pa);
#pragma unroll
    for (int g = 0; g < H; ++g) {
      vu_f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = stats[H * H + g];          // folded tables: scalar loads
#pragma unroll
        for (int h = 0; h < H; ++h) a += stats[g * H + h] * pt[h][e];
        o.v[e] = (jc + e < N) ? a : 0.f;
      }
      vu_st4(Ah + off + g * hs, o);
    }
  }
}

// backward pass 1: s1_g = sum dAhat_g ; s2_g = sum dAhat_g * xhat_g
template <typename T, int H>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ Ps, const T* __restrict__ dA,
                                                           const float* __restrict__ W, const float* __restrict__ c,
                                                           const float* __restrict__ stats, float* partials,
                                                           int B, int N, int ld, float inv_keep) {
  __shared__ float sW[H * H + H];   // W * rstd_g ; (c - mean) * rstd
  __shared__ float red[4][2 * H];
  for (int i = threadIdx.x; i < H * H; i += blockDim.x) sW[i] = W[i] * stats[H * H + 2 * H + i / H];
  for (int i = threadIdx.x; i < H; i += blockDim.x) sW[H * H + i] = (c[i] - stats[H * H + H + i]) * stats[H * H + 2 * H + i];
  __syncthreads();
  const int ld4 = ld >> 2;
  const long long total = (long long)B * N * ld4;
  const long long hs = (long long)N * ld;
  float s1[H], s2[H];
#pragma unroll
  for (int g = 0; g < H; ++g) { s1[g] = 0.f; s2[g] = 0.f; }
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    // 32-bit index arithmetic (launcher guarantees < 2^32 vector positions): 64-bit divisions by
    // run-time values cost more than the mix itself
    const unsigned t32 = (unsigned)t;
    const unsigned row = t32 / (unsigned)ld4;
    const int jc = (int)(t32 - row * (unsigned)ld4) << 2;
    const long long b = row / (unsigned)N;
    const int i = (int)(row - (unsigned)b * (unsigned)N);
    const long long off = (b * H * N + i) * (long long)ld + jc;
    float pt[H][4], pa[H][4];
    load_heads4<T, H>(Ps + off, hs, inv_keep, pt, pa);
#pragma unroll
    for (int g = 0; g < H; ++g) {
      const vu_f4 d = vu_ld4(dA + off + g * hs);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh = sW[H * H + g];
#pragma unroll
        for (int h = 0; h < H; ++h) xh += sW[g * H + h] * pt[h][e];
        if (jc + e < N) { s1[g] += d.v[e]; s2[g] += d.v[e] * xh; }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < H; ++g) {
    const float a = vu_wave_sum(s1[g]), b2 = vu_wave_sum(s2[g]);
    if (lane == 0) { red[wave][g] = a; red[wave][H + g] = b2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * H)
    partials[blockIdx.x * 2 * H + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void bn_bwd_finalize_kernel(const float* partials, int nblocks, float* stats, float* dgamma,
                                       float* dbeta, int H, double count, int training) {
  __shared__ double sd[256];
  for (int q = 0; q < 2 * H; ++q) {
    double a = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) a += (double)partials[i * 2 * H + q];
    sd[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
      const int g = q < H ? q : q - H;
      if (q < H) { dbeta[g] += (float)sd[0]; stats[H * H + 3 * H + g] = training ? (float)(sd[0] / count) : 0.f; }
      else { dgamma[g] += (float)sd[0]; stats[H * H + 4 * H + g] = training ? (float)(sd[0] / count) : 0.f; }
    }
    __syncthreads();
  }
}

// backward pass 2 (one wave per (b,i) row, all heads): dAhat -> dA -> dP~ -> dP -> dS, in place.
// Rows of any length: two sweeps over the row (the softmax backward needs the row-wide sum
// delta_h = sum_j dP_h P_h before the first dS can be written); a lane owns V consecutive columns
// per trip (V = 4, or 2 for 8 heads), loads run one trip ahead of the arithmetic, the second sweep
// re-reads a row the wave has just streamed.  (Measured on the 512x512x1 configuration, N = 4096,
// 8 heads: scalar loads 13.0 ms per launch; vector loads 9.9; two waves per SIMD 7.9; loads one
// trip ahead 6.8.  Leaving dP in HBM after sweep 1 to make sweep 2 element-wise was slower, 8.0.)
template <int V> struct vu_fv { float v[V]; };
template <int V> __device__ __forceinline__ vu_fv<V> ld_vec(const float* p) {
  vu_fv<V> r;
  if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
  else { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
  return r;
}
template <int V> __device__ __forceinline__ vu_fv<V> ld_vec(const bf16_t* p) {
  vu_fv<V> r;
  if constexpr (V == 4) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    r.v[0] = __uint_as_float(t.x << 16); r.v[1] = __uint_as_float(t.x & 0xffff0000u);
    r.v[2] = __uint_as_float(t.y << 16); r.v[3] = __uint_as_float(t.y & 0xffff0000u);
  } else {
    const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
    r.v[0] = __uint_as_float(t << 16); r.v[1] = __uint_as_float(t & 0xffff0000u);
  }
  return r;
}
template <int V> __device__ __forceinline__ void st_vec(float* p, const vu_fv<V>& a) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
  else *reinterpret_cast<float2*>(p) = make_float2(a.v[0], a.v[1]);
}
template <int V> __device__ __forceinline__ void st_vec(bf16_t* p, const vu_fv<V>& a) {
  if constexpr (V == 4) { bf16x4 t = {(bf16_t)a.v[0], (bf16_t)a.v[1], (bf16_t)a.v[2], (bf16_t)a.v[3]}; *reinterpret_cast<bf16x4*>(p) = t; }
  else { typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_; bf16x2_ t = {(bf16_t)a.v[0], (bf16_t)a.v[1]}; *reinterpret_cast<bf16x2_*>(p) = t; }
}

template <typename T, int V> struct vu_raw { uint32_t w[sizeof(T) * V / 4]; };
template <typename T, int V> __device__ __forceinline__ vu_raw<T, V> ld_raw(const T* p) {
  vu_raw<T, V> r;
  constexpr int NW = sizeof(T) * V / 4;
  if constexpr (NW == 1) r.w[0] = *reinterpret_cast<const uint32_t*>(p);
  else if constexpr (NW == 2) { const uint2 t = *reinterpret_cast<const uint2*>(p); r.w[0] = t.x; r.w[1] = t.y; }
  else { const uint4 t = *reinterpret_cast<const uint4*>(p); r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w; }
  return r;
}
template <int V> __device__ __forceinline__ vu_fv<V> cvt_raw(const vu_raw<float, V>& a) {
  vu_fv<V> r;
#pragma unroll
  for (int q = 0; q < V; ++q) r.v[q] = __uint_as_float(a.w[q]);
  return r;
}
template <int V> __device__ __forceinline__ vu_fv<V> cvt_raw(const vu_raw<bf16_t, V>& a) {
  vu_fv<V> r;
#pragma unroll
  for (int q = 0; q < V; ++q) r.v[q] = (q & 1) ? __uint_as_float(a.w[q >> 1] & 0xffff0000u) : __uint_as_float(a.w[q >> 1] << 16);
  return r;
}

template <typename T, int H>
__global__ __launch_bounds__(256, 2) void map_bwd_kernel(const T* __restrict__ Ps, T* dA, const float* __restrict__ W,
                                                      const float* __restrict__ c, const float* __restrict__ gamma,
                                                      const float* __restrict__ stats, float* dW, float* dc,
                                                      long long rows, int N, int ld, float inv_keep, float scale) {
  __shared__ float sW[H * H];      // W
  __shared__ float sX[H * H + H];  // W*rstd, (c-mean)*rstd   -> xhat
  __shared__ float sG[3 * H];      // gamma*rstd, m1, m2
  __shared__ float red[4][H * H + H];
  for (int i = threadIdx.x; i < H * H; i += blockDim.x) { sW[i] = W[i]; sX[i] = W[i] * stats[H * H + 2 * H + i / H]; }
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    sX[H * H + i] = (c[i] - stats[H * H + H + i]) * stats[H * H + 2 * H + i];
    sG[i] = gamma[i] * stats[H * H + 2 * H + i];
    sG[H + i] = stats[H * H + 3 * H + i];
    sG[2 * H + i] = stats[H * H + 4 * H + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long hs = (long long)N * ld;
  constexpr int V = H >= 8 ? 2 : 4;
  const int n4 = (N + V - 1) / V * V;   // ld is a multiple of 8 >= N: a group starting below N lies inside the row
  float aW[H * H], ac[H];
#pragma unroll
  for (int i = 0; i < H * H; ++i) aW[i] = 0.f;
#pragma unroll
  for (int i = 0; i < H; ++i) ac[i] = 0.f;
  const long long wstride = (long long)gridDim.x * 4;
  for (long long row = blockIdx.x * 4LL + wave; row < rows; row += wstride) {
    const long long b = row / N;
    const int i = (int)(row - b * N);
    const long long off = (b * H * N + i) * (long long)ld;
    float delta[H];
#pragma unroll
    for (int h = 0; h < H; ++h) delta[h] = 0.f;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      vu_raw<T, V> np_[H], nd_[H];      // the next trip's operands, loaded one trip ahead
      if (lane * V < N) {
#pragma unroll
        for (int h = 0; h < H; ++h) { np_[h] = ld_raw<T, V>(Ps + off + h * hs + lane * V); nd_[h] = ld_raw<T, V>(dA + off + h * hs + lane * V); }
      }
#pragma unroll 1
      for (int j = lane * V; j < N; j += 64 * V) {
        vu_raw<T, V> cp_[H], cd_[H];
#pragma unroll
        for (int h = 0; h < H; ++h) { cp_[h] = np_[h]; cd_[h] = nd_[h]; }
        if (j + 64 * V < N) {
#pragma unroll
          for (int h = 0; h < H; ++h) {
            np_[h] = ld_raw<T, V>(Ps + off + h * hs + j + 64 * V);
            nd_[h] = ld_raw<T, V>(dA + off + h * hs + j + 64 * V);
          }
        }
        asm volatile("" ::: "memory");   // keep the 8x8 constants in LDS: hoisted into registers they cost the second wave per SIMD
        float pt[H][V], pa[H][V], dAg[H][V];
        bool ok[V];
#pragma unroll
        for (int q = 0; q < V; ++q) ok[q] = j + q < N;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const vu_fv<V> v = cvt_raw<V>(cp_[h]);
#pragma unroll
          for (int q = 0; q < V; ++q) {
            const float x = ok[q] ? v.v[q] : 0.f;
            pa[h][q] = fabsf(x);
            pt[h][q] = x > 0.f ? x * inv_keep : 0.f;     // pt > 0 <=> kept
          }
        }
#pragma unroll
        for (int g = 0; g < H; ++g) {
          const vu_fv<V> d = cvt_raw<V>(cd_[g]);
          const float x0 = sX[H * H + g], g0 = sG[g], m1 = sG[H + g], m2 = sG[2 * H + g];
#pragma unroll
          for (int q = 0; q < V; ++q) {
            float xh = x0;
#pragma unroll
            for (int h = 0; h < H; ++h) xh += sX[g * H + h] * pt[h][q];
            dAg[g][q] = ok[q] ? g0 * (d.v[q] - m1 - xh * m2) : 0.f;
          }
        }
        if (pass == 0) {
#pragma unroll
          for (int g = 0; g < H; ++g) {
#pragma unroll
            for (int q = 0; q < V; ++q) ac[g] += dAg[g][q];
#pragma unroll
            for (int h = 0; h < H; ++h)
#pragma unroll
              for (int q = 0; q < V; ++q) aW[g * H + h] += dAg[g][q] * pt[h][q];
          }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
          vu_fv<V> o;
#pragma unroll
          for (int q = 0; q < V; ++q) {
            float dp = 0.f;
#pragma unroll
            for (int g = 0; g < H; ++g) dp += sW[g * H + h] * dAg[g][q];
            dp = pt[h][q] > 0.f ? dp * inv_keep : 0.f;
            if (pass == 0) delta[h] += dp * pa[h][q];
            else o.v[q] = pa[h][q] * (dp - delta[h]) * scale;
          }
          if (pass == 1) st_vec<V>(dA + off + h * hs + j, o);
        }
      }
      if (pass == 0) {
#pragma unroll
        for (int h = 0; h < H; ++h) delta[h] = vu_wave_sum(delta[h]);
      }
    }
    for (int j = n4 + lane; j < ld; j += 64)
#pragma unroll
      for (int h = 0; h < H; ++h) vu_st(dA + off + h * hs + j, 0.f);
  }
#pragma unroll
  for (int i = 0; i < H * H; ++i) { const float v = vu_wave_sum(aW[i]); if (lane == 0) red[wave][i] = v; }
#pragma unroll
  for (int i = 0; i < H; ++i) { const float v = vu_wave_sum(ac[i]); if (lane == 0) red[wave][H * H + i] = v; }
  __syncthreads();
  for (int i = threadIdx.x; i < H * H + H; i += blockDim.x) {
    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i < H * H) atomicAdd(dW + i, v); else atomicAdd(dc + (i - H * H), v);
  }
}

#define VU_HEADS(Hv, ...) \
  switch (Hv) { case 1: { constexpr int HH = 1; __VA_ARGS__ } break; case 2: { constexpr int HH = 2; __VA_ARGS__ } break; \
                case 4: { constexpr int HH = 4; __VA_ARGS__ } break; case 8: { constexpr int HH = 8; __VA_ARGS__ } break; \
                default: vu_set_error("num_heads %d not supported by the map kernels (1,2,4,8)", Hv); return VU_EUNSUPPORTED; }

int vu_k_mix_stats(int dtype, const void* Ps, const float* W, const float* c, float* partials, int nblocks,
                   int B, int H, int N, int ld, float inv_keep, hipStream_t st) {
  VU_REQUIRE(ld % 4 == 0, "mix_stats: ld %% 4");
  VU_REQUIRE((long long)B * N * (ld / 4) < 4294967295LL, "mix kernels: more than 2^32 vector positions");
  {
    const int mm = vu_k_mix_stats_mm(dtype, Ps, W, partials, nullptr, nblocks, B, H, N, ld, inv_keep, st);
    if (mm <= 0) return mm;
  }
  VU_HEADS(H, VU_DISPATCH_T(dtype, hipLaunchKernelGGL((mix_stats_kernel<T, HH>), dim3(nblocks), dim3(256), 0, st, (const T*)Ps, W, c, partials, B, N, ld, inv_keep);))
  if (vu_prof_on()) vu_prof_note("mix_stats_kernel", 0.0, (double)B * H * N * N * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_mix_stats");
}
int vu_k_bn_finalize(const float* partials, int nblocks, const float* W, const float* c, const float* gamma,
                     const float* beta, float* run_mean, float* run_var, float* stats, int H, int N, double count,
                     int training, float momentum, float eps, hipStream_t st) {
  VU_REQUIRE(H <= 16, "bn_finalize: H > 16");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, st, partials, nblocks, W, c, gamma, beta, run_mean, run_var, stats, H, N, count, training, momentum, eps);
  return vu_check_launch("vu_bn_finalize");
}
int vu_k_mix_apply(int dtype, const void* Ps, void* Ahat, const float* stats, int B, int H, int N, int ld,
                   float inv_keep, hipStream_t st) {
  const long long total = (long long)B * N * (ld / 4);
  const int grid = grid_for(total, 256, 256 * 32);
  VU_HEADS(H, VU_DISPATCH_T(dtype, hipLaunchKernelGGL((mix_apply_kernel<T, HH>), dim3(grid), dim3(256), 0, st, (const T*)Ps, (T*)Ahat, stats, B, N, ld, inv_keep);))
  if (vu_prof_on()) vu_prof_note("mix_apply_kernel", 0.0, (double)B * H * N * N * 2 * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_mix_apply");
}
int vu_k_bn_bwd_stats(int dtype, const void* Ps, const void* dAhat, const float* W, const float* c,
                      const float* stats, float* partials, int nblocks, int B, int H, int N, int ld,
                      float inv_keep, hipStream_t st) {
  VU_HEADS(H, VU_DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_stats_kernel<T, HH>), dim3(nblocks), dim3(256), 0, st, (const T*)Ps, (const T*)dAhat, W, c, stats, partials, B, N, ld, inv_keep);))
  if (vu_prof_on()) vu_prof_note("bn_bwd_stats_kernel", 0.0, (double)B * H * N * N * 2 * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_bn_bwd_stats");
}
int vu_k_bn_bwd_finalize(const float* partials, int nblocks, float* stats, float* dgamma, float* dbeta, int H,
                         double count, int training, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, partials, nblocks, stats, dgamma, dbeta, H, count, training);
  return vu_check_launch("vu_bn_bwd_finalize");
}
int vu_k_map_bwd_2sweep(int dtype, const void* Ps, void* dAhat_dS, const float* W, const float* c, const float* gamma,
                 const float* stats, float* dW, float* dc, int B, int H, int N, int ld, float inv_keep,
                 float scale, hipStream_t st) {
  const long long rows = (long long)B * N;
  long long grid = (rows + 3) / 4;
  if (grid > 256 * 8) grid = 256 * 8;
  VU_HEADS(H, VU_DISPATCH_T(dtype, hipLaunchKernelGGL((map_bwd_kernel<T, HH>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)Ps, (T*)dAhat_dS, W, c, gamma, stats, dW, dc, rows, N, ld, inv_keep, scale);))
  if (vu_prof_on()) vu_prof_note("map_bwd_kernel", 0.0, (double)B * H * N * N * 3 * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_map_bwd");
}

// =============================================================================================
// K13 residual + LayerNorm((N,D)) (model.py:193-196,203-206): statistics over all P = N*D
// elements of a sample; affine weight/bias of shape (N,D).  Two launches forward (chunk stats
// with Chan merge, apply), two backward.
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void add_ln_stats_kernel(const T* a, const T* __restrict__ x,
                                                           T* z, float* partials, long long P) {
  __shared__ float sm[16];
  const int c = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
  const long long base = (long long)c * VU_LN_CHUNK;
  const long long sb = (long long)b * P;
  float v[8 * VU_LN_ITS];
  int cnt = 0;
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < 2 * VU_LN_ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 4;
    if (e < P) {
      vu_f4 t = vu_ld4(a + sb + e);
      if (x || z != a) {   // (z == a, no x: the sum was already formed by the producing GEMM's epilogue)
        if (x) { const vu_f4 u = vu_ld4(x + sb + e); t.v[0] += u.v[0]; t.v[1] += u.v[1]; t.v[2] += u.v[2]; t.v[3] += u.v[3]; }
        vu_st4(z + sb + e, t);
        // statistics are taken on the values as stored (so forward and backward agree bit for bit)
        t = vu_ld4(z + sb + e);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[it * 4 + q] = t.v[q]; sum += t.v[q]; }
      cnt += 4;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[it * 4 + q] = 0.f;
    }
  }
  const float tot = vu_block_sum(sum, sm);
  long long n = P - base; if (n > VU_LN_CHUNK) n = VU_LN_CHUNK;
  const float mean = tot / (float)n;
  float m2 = 0.f;
#pragma unroll
  for (int it = 0; it < 2 * VU_LN_ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 4;
    if (e < P) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float d = v[it * 4 + q] - mean; m2 += d * d; }
    }
  }
  const float M2 = vu_block_sum(m2, sm);
  if (threadIdx.x == 0) {
    float* o = partials + ((long long)b * nch + c) * 3;
    o[0] = (float)n; o[1] = mean; o[2] = M2;
  }
}

// Pooled mean / variance of a sample from its per-chunk (count, mean, M2) triples, by the first wave
// of the block in parallel: mu = sum n_c mu_c / sum n_c ;  M2 = sum (M2_c + n_c (mu_c - mu)^2).
__device__ __forceinline__ void ln_merge_stats(const float* partials, int b, int nch, float eps, float* sm2,
                                               float& mean, float& rstd) {
  if (threadIdx.x < 64) {
    double n = 0.0, s = 0.0;
    for (int c = threadIdx.x; c < nch; c += 64) {
      const float* o = partials + ((long long)b * nch + c) * 3;
      n += (double)o[0]; s += (double)o[0] * (double)o[1];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { n += __shfl_xor(n, m, 64); s += __shfl_xor(s, m, 64); }
    const double mu = s / n;
    double M2 = 0.0;
    for (int c = threadIdx.x; c < nch; c += 64) {
      const float* o = partials + ((long long)b * nch + c) * 3;
      const double d = (double)o[1] - mu;
      M2 += (double)o[2] + (double)o[0] * d * d;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) M2 += __shfl_xor(M2, m, 64);
    if (threadIdx.x == 0) { sm2[0] = (float)mu; sm2[1] = (float)(1.0 / sqrt(M2 / n + (double)eps)); }
  }
  __syncthreads();
  mean = sm2[0]; rstd = sm2[1];
}

template <typename T>
__global__ __launch_bounds__(256) void ln_apply_kernel(const T* __restrict__ z, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ y,
                                                       const float* partials, float* stats, long long P, float eps) {
  __shared__ float sm2[2];
  const int c = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
  const long long base = (long long)c * VU_LN_CHUNK, sb = (long long)b * P;
  vu_f4 tz[2 * VU_LN_ITS];       // the chunk's values are in flight while the statistics are merged
#pragma unroll
  for (int it = 0; it < 2 * VU_LN_ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 4;
    if (e < P) tz[it] = vu_ld4(z + sb + e);
  }
  float mean, rstd;
  ln_merge_stats(partials, b, nch, eps, sm2, mean, rstd);
  if (c == 0 && threadIdx.x == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
#pragma unroll
  for (int it = 0; it < 2 * VU_LN_ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 4;
    if (e < P) {
      vu_f4 t = tz[it];
      const float4 ww = *reinterpret_cast<const float4*>(w + e), bb = *reinterpret_cast<const float4*>(bias + e);
      t.v[0] = (t.v[0] - mean) * rstd * ww.x + bb.x;
      t.v[1] = (t.v[1] - mean) * rstd * ww.y + bb.y;
      t.v[2] = (t.v[2] - mean) * rstd * ww.z + bb.z;
      t.v[3] = (t.v[3] - mean) * rstd * ww.w + bb.w;
      vu_st4(y + sb + e, t);
    }
  }
}

// bf16 storage, P % 8 == 0, 16-byte aligned rows: the same two passes with 16-byte accesses (8 elements per lane and load:
// half the memory instructions of the 4-element form; 19.3 MB tensors, 2 368 workgroups of 4 096 elements).
__device__ __forceinline__ uint4 vu_pack8(const float (&v)[8]) {
  union U8 { uint4 u; bf16_t h[8]; } t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t.h[e] = (bf16_t)v[e];
  return t.u;
}
__device__ __forceinline__ void vu_unpack8(const uint4& u, float (&v)[8]) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <int ITS>
__global__ __launch_bounds__(256) void add_ln_stats8_kernel(const bf16_t* a, const bf16_t* __restrict__ x, bf16_t* z, float* partials, long long P) {
  __shared__ float sm[16];
  const int c = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
  const long long base = (long long)c * (2048 * ITS), sb = (long long)b * P;
  float v[ITS][8];
  float sum = 0.f;
  const bool writes = x || z != a;      // (z == a, no x: the sum was already formed by the producing GEMM's epilogue)
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    if (e < P) {
      vu_unpack8(*reinterpret_cast<const uint4*>(a + sb + e), v[it]);
      if (writes) {
        if (x) {
          float u[8];
          vu_unpack8(*reinterpret_cast<const uint4*>(x + sb + e), u);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[it][q] += u[q];
        }
        const uint4 pk = vu_pack8(v[it]);
        *reinterpret_cast<uint4*>(z + sb + e) = pk;
        vu_unpack8(pk, v[it]);           // statistics on the values as stored (forward and backward agree bit for bit)
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += v[it][q];
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[it][q] = 0.f;
    }
  }
  const float tot = vu_block_sum(sum, sm);
  long long n = P - base; if (n > (2048 * ITS)) n = (2048 * ITS);
  const float mean = tot / (float)n;
  float m2 = 0.f;
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    if (e < P) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float d = v[it][q] - mean; m2 += d * d; }
    }
  }
  const float M2 = vu_block_sum(m2, sm);
  if (threadIdx.x == 0) {
    float* o = partials + ((long long)b * nch + c) * 3;
    o[0] = (float)n; o[1] = mean; o[2] = M2;
  }
}
template <int ITS>
__global__ __launch_bounds__(256) void ln_apply8_kernel(const bf16_t* __restrict__ z, const float* __restrict__ w, const float* __restrict__ bias,
                                                        bf16_t* __restrict__ y, const float* partials, float* stats, long long P, float eps) {
  __shared__ float sm2[2];
  const int c = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
  const long long base = (long long)c * (2048 * ITS), sb = (long long)b * P;
  uint4 tz[ITS];       // the chunk's values are in flight while the statistics are merged
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    tz[it] = e < P ? *reinterpret_cast<const uint4*>(z + sb + e) : make_uint4(0, 0, 0, 0);
  }
  float mean, rstd;
  ln_merge_stats(partials, b, nch, eps, sm2, mean, rstd);
  if (c == 0 && threadIdx.x == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    if (e < P) {
      float t[8];
      vu_unpack8(tz[it], t);
      const float4 w0 = *reinterpret_cast<const float4*>(w + e), w1 = *reinterpret_cast<const float4*>(w + e + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(bias + e), b1 = *reinterpret_cast<const float4*>(bias + e + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = (t[q] - mean) * rstd * wv[q] + bv[q];
      *reinterpret_cast<uint4*>(y + sb + e) = vu_pack8(t);
    }
  }
}
inline bool ln_big_chunk(int B, long long P) {
  static const int force = [] { const char* e = getenv("VU_LN_CHUNK8K"); return e ? atoi(e) : -1; }();      // A/B switch: 0 / 1
  if (force >= 0) return force != 0;
  return (long long)B * ((P + 4095) / 4096) > 2048;
}
inline bool ln_wide_ok(int dtype, long long P, const void* p0, const void* p1, const void* p2, const void* p3) {
  static const bool off = [] { const char* e = getenv("VU_LN_WIDE"); return e && e[0] == '0'; }();      // A/B switch
  return !off && dtype == 1 && P % 8 == 0 && !(((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)p3) & 15);
}

int vu_k_add_ln_fwd(int dtype, const void* a, const void* x, void* z, const float* w, const float* bias,
                    void* y, float* partials, float* stats, int B, long long P, float eps, hipStream_t st) {
  VU_REQUIRE(P % 4 == 0, "layernorm: P %% 4");
  const int nch = vu_ln_nchunks(P);
  if (ln_wide_ok(dtype, P, a, x, z, y) && !(((uintptr_t)w | (uintptr_t)bias) & 15)) {
    // chunk of 4096 elements per workgroup; 8192 once the 4096-element grid would not be resident at once (8 workgroups per CU:
    // 2368 workgroups at 64 images leave a second, nearly empty round): 19.4 -> 17.4 us forward, 29.7 -> 28.3 backward at 64 images
    // (tools/ln_time.py; at 16 images the larger chunk is 0.3 us slower)
    if (ln_big_chunk(B, P)) {
      const int nc = (int)((P + 8191) / 8192);
      hipLaunchKernelGGL(add_ln_stats8_kernel<4>, dim3(nc, B), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)x, (bf16_t*)z, partials, P);
      hipLaunchKernelGGL(ln_apply8_kernel<4>, dim3(nc, B), dim3(256), 0, st, (const bf16_t*)z, w, bias, (bf16_t*)y, partials, stats, P, eps);
    } else {
      hipLaunchKernelGGL(add_ln_stats8_kernel<2>, dim3(nch, B), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)x, (bf16_t*)z, partials, P);
      hipLaunchKernelGGL(ln_apply8_kernel<2>, dim3(nch, B), dim3(256), 0, st, (const bf16_t*)z, w, bias, (bf16_t*)y, partials, stats, P, eps);
    }
    if (vu_prof_on()) vu_prof_note("add_ln_fwd(2 kernels)", 0.0, (double)B * P * 5 * 2.0 + (double)P * 8);
    return vu_check_launch("vu_add_ln_fwd");
  }
  VU_DISPATCH_T(dtype,
    hipLaunchKernelGGL((add_ln_stats_kernel<T>), dim3(nch, B), dim3(256), 0, st, (const T*)a, (const T*)x, (T*)z, partials, P);
    hipLaunchKernelGGL((ln_apply_kernel<T>), dim3(nch, B), dim3(256), 0, st, (const T*)z, w, bias, (T*)y, partials, stats, P, eps);)
  if (vu_prof_on()) vu_prof_note("add_ln_fwd(2 kernels)", 0.0, (double)B * P * 5 * (dtype == 0 ? 4.0 : 2.0) + (double)P * 8);
  return vu_check_launch("vu_add_ln_fwd");
}

// backward A: one block per chunk of 256 elements of P.  Thread = (tx: 8 consecutive elements,
// ts: sample slice b = ts, ts+8, ...): the affine gradients of the chunk are summed over the batch
// inside the block (registers, then one LDS exchange across the 8 slices - no atomics), and the
// per-(sample,chunk) partial sums c1 = sum dy*w, c2 = sum dy*w*xhat go to partials2.
// SL sample slices per block (8: 256 threads, the default; 16: 512 threads, VU_LN_BSL=16 - built in round 6 to put more loads in
// flight per chunk (588 blocks of 4 waves at 2.9 TB/s) and measured slower: 21.9 vs 19.5 us at 32 images, 31.9 vs 28.8 at 64)
template <typename T, int SL = 8, int KU = 4>
__global__ __launch_bounds__(32 * SL) void ln_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                           const float* __restrict__ w, const float* __restrict__ stats,
                                                           float* dw, float* db, float* partials2, int B, long long P) {
  __shared__ float red[2][SL][8][33];     // [gw|gb][slice][q][tx]
  const int tx = threadIdx.x & 31, ts = threadIdx.x >> 5;
  const int c = blockIdx.x, nch = gridDim.x;
  const long long e = (long long)c * VU_LN_BCHUNK + tx * 8;
  const bool ok0 = e < P, ok1 = e + 4 < P;
  float wv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ok0) { const float4 t = *reinterpret_cast<const float4*>(w + e); wv[0] = t.x; wv[1] = t.y; wv[2] = t.z; wv[3] = t.w; }
  if (ok1) { const float4 t = *reinterpret_cast<const float4*>(w + e + 4); wv[4] = t.x; wv[5] = t.y; wv[6] = t.z; wv[7] = t.w; }
  float gw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const vu_f4 zero = {{0.f, 0.f, 0.f, 0.f}};
  // (the accumulators this thread adds into at the end: fetched now, not behind the exchange)
  const long long et0 = (long long)c * VU_LN_BCHUNK + threadIdx.x;
  float dw0 = 0.f, db0 = 0.f;
  if (threadIdx.x < 256 && et0 < P) { dw0 = dw[et0]; db0 = db[et0]; }
  for (int bb = ts; bb < B; bb += KU * SL) {        // KU samples (4 KU loads) in flight per thread
    vu_f4 d0[KU], d1[KU], z0[KU], z1[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int b = bb + SL * k;
      const long long o = (long long)b * P + e;
      const bool v0 = ok0 && b < B, v1 = ok1 && b < B;
      if (v0 && v1 && (P & 7) == 0) {        // whole 8-element chunk, 16-byte aligned rows: one load per operand
        vu_ld8(dy + o, d0[k], d1[k]); vu_ld8(z + o, z0[k], z1[k]);
      } else {
        d0[k] = v0 ? vu_ld4(dy + o) : zero; z0[k] = v0 ? vu_ld4(z + o) : zero;
        d1[k] = v1 ? vu_ld4(dy + o + 4) : zero; z1[k] = v1 ? vu_ld4(z + o + 4) : zero;
      }
    }
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int b = bb + SL * k;
      if (b < B) {
        const float mean = stats[2 * b], rstd = stats[2 * b + 1];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float dv = q < 4 ? d0[k].v[q & 3] : d1[k].v[q & 3], zv = q < 4 ? z0[k].v[q & 3] : z1[k].v[q & 3];
          const bool okq = q < 4 ? ok0 : ok1;
          const float xh = okq ? (zv - mean) * rstd : 0.f, g = dv * wv[q];
          gw[q] = fmaf(dv, xh, gw[q]); gb[q] += dv;
          c1 += g; c2 = fmaf(g, xh, c2);
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) { c1 += __shfl_xor(c1, m, 64); c2 += __shfl_xor(c2, m, 64); }
        if (tx == 0) { float* o2 = partials2 + ((long long)b * nch + c) * 2; o2[0] = c1; o2[1] = c2; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { red[0][ts][q][tx] = gw[q]; red[1][ts][q][tx] = gb[q]; }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int t = threadIdx.x, q = t & 7, x = t >> 3;       // chunk element t = x*8 + q
    float sw = 0.f, sb2 = 0.f;
#pragma unroll
    for (int k = 0; k < SL; ++k) { sw += red[0][k][q][x]; sb2 += red[1][k][q][x]; }
    const long long et = (long long)c * VU_LN_BCHUNK + t;
    if (et < P) { dw[et] = dw0 + sw; db[et] = db0 + sb2; }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                           const float* __restrict__ w, const float* __restrict__ stats,
                                                           const float* partials2, int nbch, T* __restrict__ dz,
                                                           T* __restrict__ dzd, vu_rng rng_in, long long P) {
  const vu_rng rng = vu_rng_resolve(rng_in);
  __shared__ float sm[16];
  const int c = blockIdx.x, b = blockIdx.y;
  const long long base = (long long)c * VU_LN_CHUNK, sb = (long long)b * P;
  vu_f4 td[2 * VU_LN_ITS], tz[2 * VU_LN_ITS];      // in flight while the sample's partial sums are reduced
#pragma unroll
  for (int it = 0; it < 2 * VU_LN_ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 4;
    if (e < P) { td[it] = vu_ld4(dy + sb + e); tz[it] = vu_ld4(z + sb + e); }
  }
  float a1 = 0.f, a2 = 0.f;
  for (int i = threadIdx.x; i < nbch; i += blockDim.x) {
    const float2 o = *reinterpret_cast<const float2*>(partials2 + ((long long)b * nbch + i) * 2);
    a1 += o.x; a2 += o.y;
  }
  const float c1 = vu_block_sum(a1, sm) / (float)P;
  const float c2 = vu_block_sum(a2, sm) / (float)P;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
#pragma unroll
  for (int it = 0; it < 2 * VU_LN_ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 4;
    if (e < P) {
      const vu_f4 d = td[it], zz = tz[it];
      const float4 ww = *reinterpret_cast<const float4*>(w + e);
      const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
      vu_f4 o, od;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (zz.v[q] - mean) * rstd;
        o.v[q] = rstd * (d.v[q] * wv[q] - c1 - xh * c2);
      }
      vu_st4(dz + sb + e, o);
      if (dzd) {
        // dropout backward acts on the gradient as the next consumer sees it (rounded to T)
        const vu_f4 r = vu_ld4(dz + sb + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) od.v[q] = (rng.thr == 0 || vu_keep(rng, (uint64_t)(sb + e + q))) ? r.v[q] * rng.inv_keep : 0.f;
        vu_st4(dzd + sb + e, od);
      }
    }
  }
}

template <int ITS>
__global__ __launch_bounds__(256) void ln_bwd_apply8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ z, const float* __restrict__ w,
                                                            const float* __restrict__ stats, const float* partials2, int nbch, bf16_t* __restrict__ dz,
                                                            bf16_t* __restrict__ dzd, vu_rng rng_in, long long P) {
  const vu_rng rng = vu_rng_resolve(rng_in);
  __shared__ float sm[16];
  const int c = blockIdx.x, b = blockIdx.y;
  const long long base = (long long)c * (2048 * ITS), sb = (long long)b * P;
  uint4 td[ITS], tz[ITS];      // in flight while the sample's partial sums are reduced
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    const bool ok = e < P;
    td[it] = ok ? *reinterpret_cast<const uint4*>(dy + sb + e) : make_uint4(0, 0, 0, 0);
    tz[it] = ok ? *reinterpret_cast<const uint4*>(z + sb + e) : make_uint4(0, 0, 0, 0);
  }
  float a1 = 0.f, a2 = 0.f;
  for (int i = threadIdx.x; i < nbch; i += blockDim.x) {
    const float2 o = *reinterpret_cast<const float2*>(partials2 + ((long long)b * nbch + i) * 2);
    a1 += o.x; a2 += o.y;
  }
  const float c1 = vu_block_sum(a1, sm) / (float)P;
  const float c2 = vu_block_sum(a2, sm) / (float)P;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const long long e = base + (it * 256 + threadIdx.x) * 8;
    if (e < P) {
      float d[8], zz[8], o[8];
      vu_unpack8(td[it], d); vu_unpack8(tz[it], zz);
      const float4 w0 = *reinterpret_cast<const float4*>(w + e), w1 = *reinterpret_cast<const float4*>(w + e + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float xh = (zz[q] - mean) * rstd;
        o[q] = rstd * (d[q] * wv[q] - c1 - xh * c2);
      }
      const uint4 pk = vu_pack8(o);
      *reinterpret_cast<uint4*>(dz + sb + e) = pk;
      if (dzd) {
        // dropout backward acts on the gradient as the next consumer sees it (rounded to bf16)
        float r[8];
        vu_unpack8(pk, r);
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = (rng.thr == 0 || vu_keep(rng, (uint64_t)(sb + e + q))) ? r[q] * rng.inv_keep : 0.f;
        *reinterpret_cast<uint4*>(dzd + sb + e) = vu_pack8(r);
      }
    }
  }
}

int vu_k_ln_bwd(int dtype, const void* dy, const void* z, const float* w, const float* stats, float* dw,
                float* db, float* partials2, void* dz, void* dz_drop, vu_rng rng, int B, long long P,
                hipStream_t st) {
  VU_REQUIRE(P % 4 == 0, "layernorm: P %% 4");
  const int nbch = vu_ln_nbchunks(P), nch = vu_ln_nchunks(P);
  if (ln_wide_ok(dtype, P, dy, z, dz, dz_drop) && !((uintptr_t)w & 15)) {
    static const int bsl = [] { const char* e = getenv("VU_LN_BSL"); return e ? atoi(e) : 8; }();      // A/B switch: 8 / 16 slices (measured round 6: 16 is SLOWER, 31.9 vs 28.8 us per LayerNorm backward at 64 images)
    static const int bku = [] { const char* e = getenv("VU_LN_KU"); return e ? atoi(e) : 4; }();      // A/B switch: samples in flight per thread (measured round 6: 8 is SLOWER at 64 images, 33.4 vs 27.9 us per LayerNorm backward, equal at 32 / 16)
    if (bsl == 16 && B >= 32)
      hipLaunchKernelGGL((ln_bwd_stats_kernel<bf16_t, 16>), dim3(nbch), dim3(512), 0, st, (const bf16_t*)dy, (const bf16_t*)z, w, stats, dw, db, partials2, B, P);
    else if (bku == 8 && B > 32)
      hipLaunchKernelGGL((ln_bwd_stats_kernel<bf16_t, 8, 8>), dim3(nbch), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)z, w, stats, dw, db, partials2, B, P);
    else
    hipLaunchKernelGGL((ln_bwd_stats_kernel<bf16_t>), dim3(nbch), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)z, w, stats, dw, db, partials2, B, P);
    if (ln_big_chunk(B, P))
      hipLaunchKernelGGL(ln_bwd_apply8_kernel<4>, dim3((unsigned)((P + 8191) / 8192), B), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)z, w, stats, partials2, nbch, (bf16_t*)dz, (bf16_t*)dz_drop, rng, P);
    else
      hipLaunchKernelGGL(ln_bwd_apply8_kernel<2>, dim3(nch, B), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)z, w, stats, partials2, nbch, (bf16_t*)dz, (bf16_t*)dz_drop, rng, P);
    if (vu_prof_on()) vu_prof_note("ln_bwd(2 kernels)", 0.0, (double)B * P * 5 * 2.0 + (double)P * 24);
    return vu_check_launch("vu_ln_bwd");
  }
  VU_DISPATCH_T(dtype,
    hipLaunchKernelGGL((ln_bwd_stats_kernel<T>), dim3(nbch), dim3(256), 0, st, (const T*)dy, (const T*)z, w, stats, dw, db, partials2, B, P);
    hipLaunchKernelGGL((ln_bwd_apply_kernel<T>), dim3(nch, B), dim3(256), 0, st, (const T*)dy, (const T*)z, w, stats, partials2, nbch, (T*)dz, (T*)dz_drop, rng, P);)
  if (vu_prof_on()) vu_prof_note("ln_bwd(2 kernels)", 0.0, (double)B * P * 5 * (dtype == 0 ? 4.0 : 2.0) + (double)P * 24);
  return vu_check_launch("vu_ln_bwd");
}

// =============================================================================================
// small helpers: column sums (bias gradients), dropout, loss, optimizer
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, float* out, long long rows,
                                                     int ncols, long long ld, long long rows_per_block) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const long long r0 = blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
  float a = 0.f;
  if (col < ncols)
    for (long long r = r0 + ry; r < r1; r += 4) a += vu_ld(in + r * ld + col);
  red[ry][cx] = a;
  __syncthreads();
  if (ry == 0 && col < ncols) atomicAdd(out + col, red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}
int vu_k_colsum(int dtype, const void* in, float* out, long long rows, int ncols, long long ld, hipStream_t st) {
  if (rows == 0 || ncols == 0) return VU_OK;
  const int gx = vu_cdiv(ncols, 64);
  int gy = 512 / gx; if (gy < 1) gy = 1;
  long long rpb = vu_cdiv64(rows, gy); if (rpb < 16) rpb = 16;
  gy = (int)vu_cdiv64(rows, rpb);
  VU_DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), dim3(gx, gy), dim3(256), 0, st, (const T*)in, out, rows, ncols, ld, rpb);)
  if (vu_prof_on()) vu_prof_note("colsum_kernel", 0.0, (double)rows * ncols * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_colsum");
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ in, T* __restrict__ out, long long n4, vu_rng rng_in) {
  const vu_rng rng = vu_rng_resolve(rng_in);
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4; t += (long long)gridDim.x * blockDim.x) {
    vu_f4 v = vu_ld4(in + t * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) v.v[q] = (rng.thr == 0 || vu_keep(rng, (uint64_t)(t * 4 + q))) ? v.v[q] * rng.inv_keep : 0.f;
    vu_st4(out + t * 4, v);
  }
}
int vu_k_dropout(int dtype, const void* in, void* out, long long n, vu_rng rng, hipStream_t st) {
  VU_REQUIRE(n % 4 == 0, "dropout: n %% 4");
  if (n == 0) return VU_OK;
  VU_DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const T*)in, (T*)out, n / 4, rng);)
  if (vu_prof_on()) vu_prof_note("dropout_kernel", 0.0, (double)n * 2 * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_dropout");
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long n4) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4; t += (long long)gridDim.x * blockDim.x) {
    vu_f4 x = vu_ld4(a + t * 4);
    const vu_f4 y = vu_ld4(b + t * 4);
    x.v[0] += y.v[0]; x.v[1] += y.v[1]; x.v[2] += y.v[2]; x.v[3] += y.v[3];
    vu_st4(out + t * 4, x);
  }
}
int vu_k_add(int dtype, const void* a, const void* b, void* out, long long n, hipStream_t st) {
  VU_REQUIRE(n % 4 == 0, "add: n %% 4");
  if (n == 0) return VU_OK;
  VU_DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const T*)a, (const T*)b, (T*)out, n / 4);)
  if (vu_prof_on()) vu_prof_note("add_kernel", 0.0, (double)n * 3 * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_add");
}

// MSELoss (run_denoising.py:80): loss = mean((out-target)^2) ; dout = 2 (out-target) / n * gscale
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                  float* __restrict__ d, float* partials, long long n, float k) {
  __shared__ float sm[16];
  float acc = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float e = o[i] - t[i];
    acc += e * e;
    if (d) d[i] = e * k;
  }
  const float tot = vu_block_sum(acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}
__global__ void mse_finalize_kernel(const float* partials, int nb, float* loss, double inv_n) {
  __shared__ double sd[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) a += (double)partials[i];
  sd[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *loss = (float)(sd[0] * inv_n);
}
int vu_k_mse(const float* out, const float* target, float* dout, float* loss, float* partials, long long n,
             float gscale, hipStream_t st) {
  VU_REQUIRE(n > 0, "mse: empty");
  const int nb = grid_for(n, 256, 1024);
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, st, out, target, dout, partials, n, 2.0f * gscale / (float)n);
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, st, partials, nb, loss, 1.0 / (double)n);
  return vu_check_launch("vu_mse");
}

// AdamW (torch.optim.AdamW semantics, run_denoising.py:81) over the flat parameter arena.
// hyper = {lr, beta1, beta2, eps, weight_decay} in device memory, step counter in device memory
// (incremented by the kernel's block 0 AFTER use via a separate tiny launch) so a captured
// hipGraph replays with the right bias correction and a host-updated learning rate.
__global__ void adamw_step_kernel(int* step) { *step += 1; }
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ shadow, long long n4,
                                                    const float* __restrict__ hyper, const int* __restrict__ step,
                                                    float gscale) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const float t = (float)(*step);
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ga[4] = {gg.x, gg.y, gg.z, gg.w};
    float ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float gr = ga[q] * gscale;
      pa[q] *= (1.f - lr * wd);
      ma[q] = b1 * ma[q] + (1.f - b1) * gr;
      va[q] = b2 * va[q] + (1.f - b2) * gr * gr;
      pa[q] -= step_size * ma[q] / (sqrtf(va[q]) * inv_sqrt_bc2 + eps);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
    if (shadow) {
      bf16x4 s = {(bf16_t)pa[0], (bf16_t)pa[1], (bf16_t)pa[2], (bf16_t)pa[3]};
      reinterpret_cast<bf16x4*>(shadow)[i] = s;
    }
  }
}
int vu_k_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n, const float* hyper,
               int* step, float gscale, hipStream_t st) {
  VU_REQUIRE(n % 4 == 0, "adamw: arena length must be a multiple of 4 (pad the arena)");
  hipLaunchKernelGGL(adamw_step_kernel, dim3(1), dim3(1), 0, st, step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4, 256, 256 * 8)), dim3(256), 0, st, p, g, m, v, (bf16_t*)shadow_bf16, n / 4, hyper, step, gscale);
  if (vu_prof_on()) vu_prof_note("adamw_kernel", 0.0, (double)n * 30);
  return vu_check_launch("vu_adamw");
}

__global__ void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(in)[i];
    bf16x4 s = {(bf16_t)a.x, (bf16_t)a.y, (bf16_t)a.z, (bf16_t)a.w};
    reinterpret_cast<bf16x4*>(out)[i] = s;
  }
}
int vu_k_cast_bf16(const float* in, void* out, long long n, hipStream_t st) {
  VU_REQUIRE(n % 4 == 0, "cast: n %% 4");
  if (n == 0) return VU_OK;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4, 256, 256 * 8)), dim3(256), 0, st, in, (bf16_t*)out, n / 4);
  return vu_check_launch("vu_cast_bf16");
}
// OCP e4m3 rounding in place (vit_unet_amd.h: vu_round_e4m3).  The hardware conversion does the rounding; the clamp
// in front of it makes the saturation explicit (the instruction's own overflow behaviour depends on a mode bit).
__device__ __forceinline__ float e4m3_round(float a, float b, float& rb) {
  const float ca = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f), cb = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);   // (NaN falls through fmin/fmax)
  const int w = __builtin_amdgcn_cvt_pk_fp8_f32(a != a ? a : ca, b != b ? b : cb, 0, false);
  rb = __builtin_amdgcn_cvt_f32_fp8(w, 1);
  return __builtin_amdgcn_cvt_f32_fp8(w, 0);
}
template <typename T>
__global__ __launch_bounds__(256) void round_e4m3_kernel(T* x0, T* x1, T* x2, long long n4) {
  T* x = blockIdx.y == 0 ? x0 : (blockIdx.y == 1 ? x1 : x2);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    vu_f4 t = vu_ld4(x + 4 * i);
    t.v[0] = e4m3_round(t.v[0], t.v[1], t.v[1]);
    t.v[2] = e4m3_round(t.v[2], t.v[3], t.v[3]);
    vu_st4(x + 4 * i, t);
  }
}
int vu_k_round_e4m3(int dtype, void* x0, void* x1, void* x2, long long n, hipStream_t st) {
  VU_REQUIRE(n % 4 == 0, "round_e4m3: n %% 4");
  if (n == 0) return VU_OK;
  const int arrays = x2 ? 3 : (x1 ? 2 : 1);
  VU_DISPATCH_T(dtype,
    hipLaunchKernelGGL((round_e4m3_kernel<T>), dim3(grid_for(n / 4, 256, 256 * 4), arrays), dim3(256), 0, st, (T*)x0, (T*)x1, (T*)x2, n / 4);)
  if (vu_prof_on()) vu_prof_note("round_e4m3_kernel", 0.0, (double)arrays * n * 2 * (dtype == 0 ? 4.0 : 2.0));
  return vu_check_launch("vu_round_e4m3");
}
__global__ void fill_kernel(float* p, float v, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
int vu_k_fill(float* p, float v, long long n, hipStream_t st) {
  if (n == 0) return VU_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, st, p, v, n);
  return vu_check_launch("vu_fill");
}
