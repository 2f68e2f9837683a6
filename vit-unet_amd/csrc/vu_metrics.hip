// Losses / metrics next to the model that are not on the MSE train step: Dice loss with its
// gradient (README.md:91-101; SURVEY 8 config 5 "segmentation head": sigmoid on the 1-channel
// output, builder-defined), per-image PSNR (vit_unet/torch/functions.py:7-19) and per-image mean
// SSIM (README.md:85-89 names it; no implementation in the reference: scikit-image's
// structural_similarity defaults restated - uniform 7x7 window, K1 = 0.01, K2 = 0.03, sample
// covariance, border of (win-1)/2 cropped).  All of it is HBM-bound streaming: one read of each
// operand, vector loads, two-stage deterministic reductions (no float atomics).
#include "vu_common.h"
#include "../../include/vit_unet_amd.h"

namespace {

constexpr int kMaxBlocks = 1024;

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }

// ---- Dice ---------------------------------------------------------------------------------------
// stage 1: partial sums of (p t, p, t) per block; 4 elements per thread per trip when aligned.
template <bool SIG>
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                        float* __restrict__ partials, long long n) {
  __shared__ float sm[16];
  float si = 0.f, sp = 0.f, stt = 0.f;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(z)[i], b = reinterpret_cast<const float4*>(t)[i];
    const float pa[4] = {a.x, a.y, a.z, a.w}, tb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float p = SIG ? sigmoidf_(pa[q]) : pa[q];
      si += p * tb[q]; sp += p; stt += tb[q];
    }
  }
  for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float p = SIG ? sigmoidf_(z[i]) : z[i];
    si += p * t[i]; sp += p; stt += t[i];
  }
  const float a = vu_block_sum(si, sm), b = vu_block_sum(sp, sm), c = vu_block_sum(stt, sm);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = a;
    partials[kMaxBlocks + blockIdx.x] = b;
    partials[2 * kMaxBlocks + blockIdx.x] = c;
  }
}
// stage 2: loss = 1 - (2I+1)/(S+T+1); coef = {2/den, (2I+1)/den^2} for the gradient pass
__global__ void dice_finalize_kernel(float* partials, int nb, float* loss, double smooth) {
  __shared__ double sd[3][256];
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) {
    a += (double)partials[i]; b += (double)partials[kMaxBlocks + i]; c += (double)partials[2 * kMaxBlocks + i];
  }
  sd[0][threadIdx.x] = a; sd[1][threadIdx.x] = b; sd[2][threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 3; ++q) sd[q][threadIdx.x] += sd[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double num = 2.0 * sd[0][0] + smooth, den = sd[1][0] + sd[2][0] + smooth;
    *loss = (float)(1.0 - num / den);
    partials[3 * kMaxBlocks] = (float)(2.0 / den);
    partials[3 * kMaxBlocks + 1] = (float)(num / (den * den));
  }
}
// stage 3: dL/dp = -(2 t/den - num/den^2); dL/dz = dL/dp * p (1-p)
template <bool SIG>
__global__ __launch_bounds__(256) void dice_grad_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                        float* __restrict__ dz, const float* __restrict__ coef,
                                                        long long n, float gscale) {
  const float c0 = coef[0] * gscale, c1 = coef[1] * gscale;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(z)[i], b = reinterpret_cast<const float4*>(t)[i];
    const float pa[4] = {a.x, a.y, a.z, a.w}, tb[4] = {b.x, b.y, b.z, b.w};
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float g = c1 - c0 * tb[q];
      if (SIG) { const float p = sigmoidf_(pa[q]); r[q] = g * p * (1.f - p); } else r[q] = g;
    }
    reinterpret_cast<float4*>(dz)[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
  for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float g = c1 - c0 * t[i];
    if (SIG) { const float p = sigmoidf_(z[i]); dz[i] = g * p * (1.f - p); } else dz[i] = g;
  }
}

// ---- PSNR ---------------------------------------------------------------------------------------
// grid (chunks, B): partial sum of squared differences of one image chunk
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ partials, long long P) {
  __shared__ float sm[16];
  const float* pa = a + (long long)blockIdx.y * P;
  const float* pb = b + (long long)blockIdx.y * P;
  float acc = 0.f;
  const bool vec = (P & 3) == 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (vec) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (P >> 2); i += stride) {
      const float4 x = reinterpret_cast<const float4*>(pa)[i], y = reinterpret_cast<const float4*>(pb)[i];
      const float e0 = x.x - y.x, e1 = x.y - y.y, e2 = x.z - y.z, e3 = x.w - y.w;
      acc += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
  } else {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P; i += stride) {
      const float e = pa[i] - pb[i];
      acc += e * e;
    }
  }
  const float tot = vu_block_sum(acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}
__global__ void psnr_finalize_kernel(const float* partials, int chunks, float* out, double inv_P, double r2) {
  // one wave per image
  double a = 0.0;
  for (int i = threadIdx.x; i < chunks; i += 64) a += (double)partials[blockIdx.x * chunks + i];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(10.0 * log10(r2 / (a * inv_P)));
}

// ---- SSIM ---------------------------------------------------------------------------------------
// One block = a 16x16 tile of window positions of one (image, channel) plane; the (16+win-1)^2
// pixels of both operands are staged in LDS once, every thread sums its win x win window (five
// moments) from LDS.  win <= 11.  partials[(b*C + c)*tiles + tile] = sum of S over the tile.
constexpr int kT = 16, kMaxWin = 11, kLd = kT + kMaxWin - 1;
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                   float* __restrict__ partials, int H, int W, int win,
                                                   float c1, float c2) {
  __shared__ float sx[kLd][kLd + 1], sy[kLd][kLd + 1];
  __shared__ float sm[16];
  const int oh = H - win + 1, ow = W - win + 1;
  const int tx_n = (ow + kT - 1) / kT;
  const int ty0 = (blockIdx.x / tx_n) * kT, tx0 = (blockIdx.x % tx_n) * kT;
  const long long plane = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * H * W;
  const int span = kT + win - 1;
  for (int i = threadIdx.x; i < span * span; i += 256) {
    const int r = i / span, c = i % span, gy = ty0 + r, gx = tx0 + c;
    const bool in = gy < H && gx < W;
    sx[r][c] = in ? X[plane + (long long)gy * W + gx] : 0.f;
    sy[r][c] = in ? Y[plane + (long long)gy * W + gx] : 0.f;
  }
  __syncthreads();
  const int ly = threadIdx.x / kT, lx = threadIdx.x % kT;
  float S = 0.f;
  if (ty0 + ly < oh && tx0 + lx < ow) {
    float ux = 0.f, uy = 0.f, uxx = 0.f, uyy = 0.f, uxy = 0.f;
    for (int r = 0; r < win; ++r)
      for (int c = 0; c < win; ++c) {
        const float a = sx[ly + r][lx + c], b = sy[ly + r][lx + c];
        ux += a; uy += b; uxx += a * a; uyy += b * b; uxy += a * b;
      }
    const float np_ = (float)(win * win), inv = 1.f / np_, cov = np_ / (np_ - 1.f);
    ux *= inv; uy *= inv; uxx *= inv; uyy *= inv; uxy *= inv;
    const float vx = cov * (uxx - ux * ux), vy = cov * (uyy - uy * uy), vxy = cov * (uxy - ux * uy);
    S = ((2.f * ux * uy + c1) * (2.f * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
  }
  const float tot = vu_block_sum(S, sm);
  if (threadIdx.x == 0)
    partials[((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
}
__global__ void ssim_finalize_kernel(const float* partials, int per_image, float* out, double inv_count) {
  __shared__ double sd[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < per_image; i += blockDim.x) a += (double)partials[(long long)blockIdx.x * per_image + i];
  sd[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(sd[0] * inv_count);
}

inline int blocks_for(long long n4) {
  long long g = (n4 + 255) / 256;
  return (int)(g < 1 ? 1 : (g > kMaxBlocks ? kMaxBlocks : g));
}
constexpr int kPsnrChunks = 16;

}  // namespace

extern "C" {

size_t vu_dice_partials_floats(void) { return 3 * kMaxBlocks + 2; }

int vu_dice_loss(const float* logits, const float* target, float* dlogits, float* loss, float* partials,
                 long long n, int apply_sigmoid, float grad_scale, void* stream) {
  VU_REQUIRE(n > 0, "dice: empty");
  VU_REQUIRE(logits && target && loss && partials, "dice: null pointer");
  VU_REQUIRE((((uintptr_t)logits | (uintptr_t)target | (uintptr_t)dlogits) & 15) == 0, "dice: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nb = blocks_for(n / 4 + 1);
  if (apply_sigmoid) hipLaunchKernelGGL(dice_sums_kernel<true>, dim3(nb), dim3(256), 0, st, logits, target, partials, n);
  else hipLaunchKernelGGL(dice_sums_kernel<false>, dim3(nb), dim3(256), 0, st, logits, target, partials, n);
  if (vu_prof_on()) vu_prof_note("dice_sums_kernel", 0.0, (double)n * 8);
  int rc = vu_check_launch("vu_dice_loss/sums");
  if (rc) return rc;
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, st, partials, nb, loss, 1.0);
  if (dlogits) {
    const float* coef = partials + 3 * kMaxBlocks;
    if (apply_sigmoid) hipLaunchKernelGGL(dice_grad_kernel<true>, dim3(nb), dim3(256), 0, st, logits, target, dlogits, coef, n, grad_scale);
    else hipLaunchKernelGGL(dice_grad_kernel<false>, dim3(nb), dim3(256), 0, st, logits, target, dlogits, coef, n, grad_scale);
    if (vu_prof_on()) vu_prof_note("dice_grad_kernel", 0.0, (double)n * 12);
  }
  return vu_check_launch("vu_dice_loss");
}

size_t vu_psnr_partials_floats(int B) { return (size_t)(B > 0 ? B : 0) * kPsnrChunks; }

int vu_psnr(const float* target, const float* out, float* psnr, float* partials, int B, long long P,
            float data_range, void* stream) {
  VU_REQUIRE(B > 0 && P > 0, "psnr: empty batch");
  VU_REQUIRE(B <= 65535, "psnr: at most 65535 images per call");
  VU_REQUIRE(target && out && psnr && partials, "psnr: null pointer");
  VU_REQUIRE((P & 3) != 0 || ((((uintptr_t)target | (uintptr_t)out) & 15) == 0), "psnr: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqdiff_kernel, dim3(kPsnrChunks, B), dim3(256), 0, st, target, out, partials, P);
  if (vu_prof_on()) vu_prof_note("sqdiff_kernel", 0.0, (double)B * P * 8);
  int rc = vu_check_launch("vu_psnr/sqdiff");
  if (rc) return rc;
  hipLaunchKernelGGL(psnr_finalize_kernel, dim3(B), dim3(64), 0, st, partials, kPsnrChunks, psnr, 1.0 / (double)P,
                     (double)data_range * (double)data_range);
  return vu_check_launch("vu_psnr");
}

size_t vu_ssim_partials_floats(int B, int C, int H, int W, int win) {
  if (B <= 0 || C <= 0 || win < 3 || H < win || W < win) return 0;
  const long long tiles = (long long)((H - win + 1 + kT - 1) / kT) * ((W - win + 1 + kT - 1) / kT);
  return (size_t)(tiles * C * B);
}

int vu_ssim(const float* target, const float* out, float* ssim, float* partials, int B, int C, int H, int W,
            int win, float data_range, void* stream) {
  VU_REQUIRE(B > 0 && C > 0, "ssim: empty batch");
  VU_REQUIRE(win >= 3 && win <= kMaxWin && (win & 1), "ssim: window must be odd, 3..11 (got %d)", win);
  VU_REQUIRE(H >= win && W >= win, "ssim: window %d exceeds the image (%d x %d)", win, H, W);
  VU_REQUIRE(B <= 65535 && C <= 65535, "ssim: at most 65535 images / channels per call");
  VU_REQUIRE(target && out && ssim && partials, "ssim: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int oh = H - win + 1, ow = W - win + 1;
  const int tiles = ((oh + kT - 1) / kT) * ((ow + kT - 1) / kT);
  const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
  hipLaunchKernelGGL(ssim_kernel, dim3(tiles, C, B), dim3(256), 0, st, target, out, partials, H, W, win, c1, c2);
  if (vu_prof_on()) vu_prof_note("ssim_kernel", 0.0, (double)B * C * H * W * 8);
  int rc = vu_check_launch("vu_ssim/tiles");
  if (rc) return rc;
  hipLaunchKernelGGL(ssim_finalize_kernel, dim3(B), dim3(256), 0, st, partials, tiles * C, ssim,
                     1.0 / ((double)C * oh * ow));
  return vu_check_launch("vu_ssim");
}

}  // extern "C"
