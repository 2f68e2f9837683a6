// Device-side input pipeline of the denoising runs: what DenoisingDataset.__getitem__
// (vit_unet/torch/dataset.py:52-71) and the albumentations transforms of run_denoising.py:52-59 do
// per image on the host, done per batch on the GPU from the decoded uint8 HWC pixels:
//   cv2.resize(im, im)  ->  ShiftScaleRotate (cv2.warpAffine, bilinear for the noisy image, nearest
//   for the clean "mask", constant border 0)  ->  Normalize(mean, std, max 255) on the noisy image
//   only  ->  /255 on both  ->  HWC -> CHW float32.
// Byte / integer work, HBM bound: every source byte is read once (taps come from L2), every output
// float is written once in plane order (coalesced along x).  The arithmetic is the integer
// fixed-point scheme OpenCV publishes for 8-bit images (resize: 11-bit coefficients, two passes;
// warp: 10-bit coordinates, 1/32 sub-pixel), restated - OpenCV itself is not in the reference tree
// nor in this image, so parity with it is unpinned; parity with oracle/ is bit-exact.
#include "vu_common.h"
#include "../../include/vit_unet_amd.h"

namespace {

// cv2 resize, INTER_LINEAR, 8-bit: tap index and 11-bit coefficient pair for output index d.
__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int& s0, int& s1, int& a0, int& a1) {
  float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
  int s = (int)floorf(f);
  f = __fsub_rn(f, (float)s);
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  s0 = s;
  s1 = min(s + 1, ssize - 1);
  a0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
  a1 = __float2int_rn(__fmul_rn(f, 2048.f));
}

// one thread = one output pixel (3 or 1 channels); grid.y = image
template <int CH>
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                        int Hs, int Ws, int im, double sy, double sx, int area2) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= im * im) return;
  const int oy = p / im, ox = p % im;
  const uint8_t* s = src + (size_t)blockIdx.y * Hs * Ws * CH;
  uint8_t* d = dst + ((size_t)blockIdx.y * im * im + p) * CH;
  if (area2) {   // exact 2x reduction: OpenCV switches INTER_LINEAR to the 2x2 box mean
    const uint8_t* r0 = s + ((size_t)(2 * oy) * Ws + 2 * ox) * CH;
    const uint8_t* r1 = r0 + (size_t)Ws * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) d[c] = (uint8_t)((r0[c] + r0[CH + c] + r1[c] + r1[CH + c] + 2) >> 2);
    return;
  }
  int x0, x1, a0, a1, y0, y1, b0, b1;
  lin_coef(ox, sx, Ws, x0, x1, a0, a1);
  lin_coef(oy, sy, Hs, y0, y1, b0, b1);
  const uint8_t* r0 = s + (size_t)y0 * Ws * CH;
  const uint8_t* r1 = s + (size_t)y1 * Ws * CH;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int S0 = r0[x0 * CH + c] * a0 + r0[x1 * CH + c] * a1;
    const int S1 = r1[x0 * CH + c] * a0 + r1[x1 * CH + c] * a1;
    const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    d[c] = (uint8_t)min(max(v, 0), 255);
  }
}

// one thread = one output pixel of one image: inverse-affine sample of the resized noisy (bilinear)
// and clean (nearest) images, normalisation, CHW store.  minv == nullptr: no warp (val transform).
template <int CH>
__global__ __launch_bounds__(256) void warp_norm_kernel(const uint8_t* __restrict__ noisy, const uint8_t* __restrict__ clean,
                                                        float* __restrict__ x, float* __restrict__ y,
                                                        const double* __restrict__ minv, int im, float m255, float rden) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= im * im) return;
  const int oy = p / im, ox = p % im, b = blockIdx.y;
  const uint8_t* n = noisy ? noisy + (size_t)b * im * im * CH : nullptr;
  const uint8_t* c = clean ? clean + (size_t)b * im * im * CH : nullptr;
  int vn[CH], vc[CH];
  if (minv) {
    const double* M = minv + (size_t)b * 6;
    const double bx = __dmul_rn(__dadd_rn(__dmul_rn(M[1], (double)oy), M[2]), 1024.0);
    const double by = __dmul_rn(__dadd_rn(__dmul_rn(M[4], (double)oy), M[5]), 1024.0);
    const int adx = __double2int_rn(__dmul_rn(__dmul_rn(M[0], (double)ox), 1024.0));
    const int ady = __double2int_rn(__dmul_rn(__dmul_rn(M[3], (double)ox), 1024.0));
    const int X0 = __double2int_rn(bx), Y0 = __double2int_rn(by);
    // bilinear: 1/32 sub-pixel, weights (32-ax)(32-ay) ... sum 1024, taps outside the image are 0
    const int X = (X0 + 16 + adx) >> 5, Y = (Y0 + 16 + ady) >> 5;
    const int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
    const int w00 = (32 - ax) * (32 - ay), w01 = ax * (32 - ay), w10 = (32 - ax) * ay, w11 = ax * ay;
    const bool inx0 = sx >= 0 && sx < im, inx1 = sx + 1 >= 0 && sx + 1 < im;
    const bool iny0 = sy >= 0 && sy < im, iny1 = sy + 1 >= 0 && sy + 1 < im;
    const int xn = (X0 + 512 + adx) >> 10, yn = (Y0 + 512 + ady) >> 10;
    const bool inn = xn >= 0 && xn < im && yn >= 0 && yn < im;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (n) {
        const int p00 = (inx0 && iny0) ? n[((size_t)sy * im + sx) * CH + k] : 0;
        const int p01 = (inx1 && iny0) ? n[((size_t)sy * im + sx + 1) * CH + k] : 0;
        const int p10 = (inx0 && iny1) ? n[((size_t)(sy + 1) * im + sx) * CH + k] : 0;
        const int p11 = (inx1 && iny1) ? n[((size_t)(sy + 1) * im + sx + 1) * CH + k] : 0;
        vn[k] = (w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + 512) >> 10;
      }
      if (c) vc[k] = inn ? c[((size_t)yn * im + xn) * CH + k] : 0;
    }
  } else {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (n) vn[k] = n[(size_t)p * CH + k];
      if (c) vc[k] = c[(size_t)p * CH + k];
    }
  }
  const size_t plane = (size_t)im * im;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    // Normalize: (v - mean*255) * (1/(std*255)) in float32, then /255 (dataset.py:66)
    if (n) x[((size_t)b * CH + k) * plane + p] = __fdiv_rn(__fmul_rn(__fsub_rn((float)vn[k], m255), rden), 255.f);
    // the clean image is divided as uint8/255. (float64) and cast to float32 by unpack (dataset.py:66,80)
    if (c) y[((size_t)b * CH + k) * plane + p] = (float)__ddiv_rn((double)vc[k], 255.0);
  }
}

// ---- SegmentationDataset (vit_unet/torch/dataset.py:18-38): DICOM slice (16-bit signed) + label mask ----
// cv2 resize, INTER_LINEAR, non-8-bit types: float coefficient pair for output index d.
__device__ __forceinline__ void lin_coef_f(int d, double scale, int ssize, int& s0, int& s1, float& a0, float& a1) {
  float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
  int s = (int)floorf(f);
  f = __fsub_rn(f, (float)s);
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  s0 = s;
  s1 = min(s + 1, ssize - 1);
  a0 = __fsub_rn(1.f, f);
  a1 = f;
}

__device__ __forceinline__ int sat_s16(float v) { return min(max(__float2int_rn(v), -32768), 32767); }

// one thread = one output pixel: bilinear (float) resize of the slice, nearest resize of the mask
__global__ __launch_bounds__(256) void resize_seg_kernel(const int16_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                         int16_t* __restrict__ dimg, uint8_t* __restrict__ dmask, int Hs, int Ws,
                                                         int oh, int ow, double sy, double sx, int area2) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= oh * ow) return;
  const int oy = p / ow, ox = p % ow;
  const size_t so = (size_t)blockIdx.y * Hs * Ws, dof = (size_t)blockIdx.y * oh * ow + p;
  if (img) {
    const int16_t* s = img + so;
    if (area2) {
      const int16_t* r0 = s + (size_t)(2 * oy) * Ws + 2 * ox;
      dimg[dof] = (int16_t)(((int)r0[0] + r0[1] + r0[Ws] + r0[Ws + 1] + 2) >> 2);
    } else {
      int x0, x1, y0, y1;
      float a0, a1, b0, b1;
      lin_coef_f(ox, sx, Ws, x0, x1, a0, a1);
      lin_coef_f(oy, sy, Hs, y0, y1, b0, b1);
      const int16_t* r0 = s + (size_t)y0 * Ws;
      const int16_t* r1 = s + (size_t)y1 * Ws;
      const float S0 = __fadd_rn(__fmul_rn((float)r0[x0], a0), __fmul_rn((float)r0[x1], a1));
      const float S1 = __fadd_rn(__fmul_rn((float)r1[x0], a0), __fmul_rn((float)r1[x1], a1));
      dimg[dof] = (int16_t)sat_s16(__fadd_rn(__fmul_rn(S0, b0), __fmul_rn(S1, b1)));
    }
  }
  if (mask) {
    const int my = min((int)floor(__dmul_rn((double)oy, sy)), Hs - 1), mx = min((int)floor(__dmul_rn((double)ox, sx)), Ws - 1);
    dmask[dof] = mask[so + (size_t)my * Ws + mx];
  }
}

// one thread = one output pixel: inverse-affine sample (bilinear float weights at 1/32 sub-pixel for the
// slice, nearest for the mask, constant border 0), intensity window -> [0,1], label smoothing.
__global__ __launch_bounds__(256) void warp_seg_kernel(const int16_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                       float* __restrict__ x, float* __restrict__ y,
                                                       const double* __restrict__ minv, int oh, int ow, float lo, float range,
                                                       float keep, float floor_) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= oh * ow) return;
  const int oy = p / ow, ox = p % ow, b = blockIdx.y;
  const size_t base = (size_t)b * oh * ow;
  int vi = 0, vm = 0;
  if (minv) {
    const double* M = minv + (size_t)b * 6;
    const double bx = __dmul_rn(__dadd_rn(__dmul_rn(M[1], (double)oy), M[2]), 1024.0);
    const double by = __dmul_rn(__dadd_rn(__dmul_rn(M[4], (double)oy), M[5]), 1024.0);
    const int adx = __double2int_rn(__dmul_rn(__dmul_rn(M[0], (double)ox), 1024.0));
    const int ady = __double2int_rn(__dmul_rn(__dmul_rn(M[3], (double)ox), 1024.0));
    const int X0 = __double2int_rn(bx), Y0 = __double2int_rn(by);
    if (img) {
      const int X = (X0 + 16 + adx) >> 5, Y = (Y0 + 16 + ady) >> 5;
      const int sx = X >> 5, sy = Y >> 5;
      const float fx = __fmul_rn((float)(X & 31), 0.03125f), fy = __fmul_rn((float)(Y & 31), 0.03125f);
      const float gx = __fsub_rn(1.f, fx), gy = __fsub_rn(1.f, fy);
      const bool inx0 = sx >= 0 && sx < ow, inx1 = sx + 1 >= 0 && sx + 1 < ow;
      const bool iny0 = sy >= 0 && sy < oh, iny1 = sy + 1 >= 0 && sy + 1 < oh;
      const int16_t* s = img + base;
      const float p00 = (inx0 && iny0) ? (float)s[(size_t)sy * ow + sx] : 0.f;
      const float p01 = (inx1 && iny0) ? (float)s[(size_t)sy * ow + sx + 1] : 0.f;
      const float p10 = (inx0 && iny1) ? (float)s[(size_t)(sy + 1) * ow + sx] : 0.f;
      const float p11 = (inx1 && iny1) ? (float)s[(size_t)(sy + 1) * ow + sx + 1] : 0.f;
      float acc = __fmul_rn(p00, __fmul_rn(gy, gx));
      acc = __fadd_rn(acc, __fmul_rn(p01, __fmul_rn(gy, fx)));
      acc = __fadd_rn(acc, __fmul_rn(p10, __fmul_rn(fy, gx)));
      acc = __fadd_rn(acc, __fmul_rn(p11, __fmul_rn(fy, fx)));
      vi = sat_s16(acc);
    }
    if (mask) {
      const int xn = (X0 + 512 + adx) >> 10, yn = (Y0 + 512 + ady) >> 10;
      vm = (xn >= 0 && xn < ow && yn >= 0 && yn < oh) ? mask[base + (size_t)yn * ow + xn] : 0;
    }
  } else {
    if (img) vi = img[base + p];
    if (mask) vm = mask[base + p];
  }
  if (img) x[base + p] = fminf(fmaxf(__fdiv_rn(__fsub_rn((float)vi, lo), range), 0.f), 1.f);
  if (mask) y[base + p] = __fadd_rn(__fmul_rn((float)vm, keep), floor_);
}

}  // namespace

extern "C" {

size_t vu_denoise_prepare_scratch_bytes(int B, int im, int channels) {
  if (B <= 0 || im <= 0 || channels <= 0) return 0;
  return (size_t)2 * B * im * im * channels;
}

int vu_denoise_prepare(const uint8_t* noisy, const uint8_t* clean, float* x, float* y, uint8_t* scratch,
                       size_t scratch_bytes, const double* inv_affine, int B, int Hs, int Ws, int channels, int im,
                       float mean, float std, void* stream) {
  VU_REQUIRE(B > 0 && B <= 65535, "denoise_prepare: batch must be 1..65535 (got %d)", B);
  VU_REQUIRE(channels == 3 || channels == 1, "denoise_prepare: 1 or 3 channels (got %d)", channels);
  VU_REQUIRE(Hs > 0 && Ws > 0 && im > 0, "denoise_prepare: empty image");
  VU_REQUIRE((noisy && x) || (clean && y), "denoise_prepare: nothing to do (no noisy/x and no clean/y pair)");
  VU_REQUIRE((noisy == nullptr) == (x == nullptr) && (clean == nullptr) == (y == nullptr),
             "denoise_prepare: input and output of a pair must both be given");
  VU_REQUIRE(std > 0.f, "denoise_prepare: std must be positive");
  const bool need_resize = Hs != im || Ws != im;
  const size_t one = (size_t)B * im * im * channels;
  VU_REQUIRE(!need_resize || (scratch && scratch_bytes >= 2 * one), "denoise_prepare: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(vu_cdiv((long long)im * im, 256), B);
  const uint8_t* rn = noisy;
  const uint8_t* rc = clean;
  if (need_resize) {
    const double sy = (double)Hs / im, sx = (double)Ws / im;
    const int area2 = (Hs == 2 * im && Ws == 2 * im) ? 1 : 0;
    for (int which = 0; which < 2; ++which) {
      const uint8_t* s = which ? clean : noisy;
      if (!s) continue;
      uint8_t* d = scratch + which * one;
      if (channels == 3) hipLaunchKernelGGL(resize_u8_kernel<3>, grid, dim3(256), 0, st, s, d, Hs, Ws, im, sy, sx, area2);
      else hipLaunchKernelGGL(resize_u8_kernel<1>, grid, dim3(256), 0, st, s, d, Hs, Ws, im, sy, sx, area2);
      if (vu_prof_on()) vu_prof_note("resize_u8_kernel", 0.0, (double)B * channels * ((double)Hs * Ws + (double)im * im));
      int rc_ = vu_check_launch("vu_denoise_prepare/resize");
      if (rc_) return rc_;
      if (which) rc = d; else rn = d;
    }
  }
  const float m255 = mean * 255.0f, rden = 1.0f / (std * 255.0f);
  if (channels == 3) hipLaunchKernelGGL(warp_norm_kernel<3>, grid, dim3(256), 0, st, rn, rc, x, y, inv_affine, im, m255, rden);
  else hipLaunchKernelGGL(warp_norm_kernel<1>, grid, dim3(256), 0, st, rn, rc, x, y, inv_affine, im, m255, rden);
  if (vu_prof_on()) vu_prof_note("warp_norm_kernel", 0.0, (double)one * ((noisy ? 5.0 : 0.0) + (clean ? 5.0 : 0.0)));
  return vu_check_launch("vu_denoise_prepare");
}

size_t vu_seg_prepare_scratch_bytes(int B, int oh, int ow) {
  if (B <= 0 || oh <= 0 || ow <= 0) return 0;
  return (size_t)3 * B * oh * ow + 2;
}

int vu_seg_prepare(const int16_t* image, const uint8_t* mask, float* x, float* y, uint8_t* scratch, size_t scratch_bytes,
                   const double* inv_affine, int B, int Hs, int Ws, int oh, int ow, float lo, float hi, float ls,
                   void* stream) {
  VU_REQUIRE(B > 0 && B <= 65535, "seg_prepare: batch must be 1..65535 (got %d)", B);
  VU_REQUIRE(Hs > 0 && Ws > 0 && oh > 0 && ow > 0, "seg_prepare: empty image");
  VU_REQUIRE((image && x) || (mask && y), "seg_prepare: nothing to do (no image/x and no mask/y pair)");
  VU_REQUIRE((image == nullptr) == (x == nullptr) && (mask == nullptr) == (y == nullptr),
             "seg_prepare: input and output of a pair must both be given");
  VU_REQUIRE(hi > lo, "seg_prepare: intensity window must have hi > lo (got %g..%g)", (double)lo, (double)hi);
  VU_REQUIRE(ls >= 0.f && ls < 1.f, "seg_prepare: label smoothing must be in [0,1) (got %g)", (double)ls);
  const bool need_resize = Hs != oh || Ws != ow;
  const size_t one = (size_t)B * oh * ow;
  VU_REQUIRE(!need_resize || (scratch && scratch_bytes >= 3 * one + 2 && ((uintptr_t)scratch & 1) == 0),
             "seg_prepare: scratch too small or misaligned");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(vu_cdiv((long long)oh * ow, 256), B);
  const int16_t* ri = image;
  const uint8_t* rm = mask;
  if (need_resize) {
    // cv2: inv_scale = dsize / ssize, scale = 1 / inv_scale (both double)
    const double sy = 1.0 / ((double)oh / Hs), sx = 1.0 / ((double)ow / Ws);
    const int area2 = (Hs == 2 * oh && Ws == 2 * ow) ? 1 : 0;
    int16_t* di = (int16_t*)scratch;
    uint8_t* dm = scratch + 2 * one;
    hipLaunchKernelGGL(resize_seg_kernel, grid, dim3(256), 0, st, image, mask, di, dm, Hs, Ws, oh, ow, sy, sx, area2);
    if (vu_prof_on()) vu_prof_note("resize_seg_kernel", 0.0, (double)B * ((double)Hs * Ws + (double)oh * ow) * ((image ? 2 : 0) + (mask ? 1 : 0)));
    int rc_ = vu_check_launch("vu_seg_prepare/resize");
    if (rc_) return rc_;
    if (image) ri = di;
    if (mask) rm = dm;
  }
  hipLaunchKernelGGL(warp_seg_kernel, grid, dim3(256), 0, st, ri, rm, x, y, inv_affine, oh, ow, lo, hi - lo, 1.0f - ls, 0.5f * ls);
  if (vu_prof_on()) vu_prof_note("warp_seg_kernel", 0.0, (double)one * ((image ? 6.0 : 0.0) + (mask ? 5.0 : 0.0)));
  return vu_check_launch("vu_seg_prepare");
}

}  // extern "C"
