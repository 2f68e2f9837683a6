"""Device-side input pipeline of the denoising runs (reference: vit_unet/torch/dataset.py:44-73 and
the albumentations transforms of run_denoising.py:52-59).

The reference decodes, resizes, augments and normalises every image on the host
(`DenoisingDataset.__getitem__`, two DataLoader workers); at the GPU path's step rate that is the
bottleneck.  Here the host only decodes (PNG -> uint8 HWC, out of scope) and draws the augmentation
parameters; `DenoisingBatchTransform` runs resize -> ShiftScaleRotate -> Normalize -> /255 -> CHW
for the whole batch with the HIP kernels of csrc/vu_data.hip (C ABI `vu_denoise_prepare`) and
returns the `{'x','y'}` dict `ImageFitter.unpack` expects.  No CPU fallback.
"""
from __future__ import annotations

import math
import random
from typing import Dict, Optional

import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr
from .fitter import ImageFitter  # noqa: F401  (the reference keeps ImageFitter in this module)


def shift_scale_rotate_matrices(B: int, im: int, shift_limit: float = 0.2, scale_limit: float = 0.2,
                                rotate_limit: float = 20.0, rng: Optional[random.Random] = None) -> np.ndarray:
    """Per-image forward 2x3 matrices of albumentations.ShiftScaleRotate(p=1.0) as
    run_denoising.py:53 configures it: angle ~ U(-rotate_limit, rotate_limit) degrees,
    scale ~ U(1 - scale_limit, 1 + scale_limit), dx, dy ~ U(-shift_limit, shift_limit) (fractions of
    the image size); rotation about (w/2 - 0.5, h/2 - 0.5)."""
    rng = rng or random
    out = np.zeros((B, 2, 3), dtype=np.float64)
    c = im / 2.0 - 0.5
    for b in range(B):
        angle = rng.uniform(-rotate_limit, rotate_limit)
        scale = rng.uniform(1.0 - scale_limit, 1.0 + scale_limit)
        dx, dy = rng.uniform(-shift_limit, shift_limit), rng.uniform(-shift_limit, shift_limit)
        a = scale * math.cos(math.radians(angle))
        s = scale * math.sin(math.radians(angle))
        out[b] = [[a, s, (1 - a) * c - s * c + dx * im], [-s, a, s * c + (1 - a) * c + dy * im]]
    return out


def invert_affine(fwd: np.ndarray) -> np.ndarray:
    """(B,2,3) forward matrices -> (B,2,3) inverse matrices, the way cv2.warpAffine inverts them."""
    M = np.asarray(fwd, dtype=np.float64).reshape(-1, 2, 3)
    out = np.empty_like(M)
    for b in range(M.shape[0]):
        m = M[b]
        D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
        D = 1.0 / D if D != 0 else 0.0
        i00, i11 = m[1, 1] * D, m[0, 0] * D
        i01, i10 = m[0, 1] * (-D), m[1, 0] * (-D)
        out[b] = [[i00, i01, -i00 * m[0, 2] - i01 * m[1, 2]], [i10, i11, -i10 * m[0, 2] - i11 * m[1, 2]]]
    return out


class DenoisingBatchTransform:
    """`train=True`: run_denoising.py:52-55 (ShiftScaleRotate + Normalize); `train=False`: :57-59
    (Normalize only).  Call with decoded uint8 batches (B,H,W,C) (numpy or torch, host or device);
    returns {'x': (B,C,im,im) float32, 'y': ...} on the device."""

    def __init__(self, im_size: int = 224, train: bool = True, mean: float = 0.456, std: float = 0.224,
                 device="cuda", seed: Optional[int] = None):
        self.im_size, self.train, self.mean, self.std, self.device = int(im_size), bool(train), float(mean), float(std), device
        self.rng = random.Random(seed)

    def _u8(self, a) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        assert t.dtype == torch.uint8 and t.dim() == 4, "expected a (B,H,W,C) uint8 batch"
        return t.to(self.device, non_blocking=True).contiguous()

    def __call__(self, noisy, clean, matrices: Optional[np.ndarray] = None) -> Dict[str, torch.Tensor]:
        n, c = self._u8(noisy), self._u8(clean)
        assert n.shape == c.shape, f"noisy {tuple(n.shape)} and clean {tuple(c.shape)} differ"
        B, H, W, Cn = n.shape
        im = self.im_size
        L = lib()
        x = torch.empty(B, Cn, im, im, dtype=torch.float32, device=n.device)
        y = torch.empty_like(x)
        nb = L.vu_denoise_prepare_scratch_bytes(B, im, Cn)
        scratch = torch.empty(max(nb, 1), dtype=torch.uint8, device=n.device)
        minv = None
        if self.train:
            fwd = matrices if matrices is not None else shift_scale_rotate_matrices(B, im, rng=self.rng)
            assert np.asarray(fwd).shape == (B, 2, 3), "matrices must be (B,2,3)"
            minv = torch.from_numpy(invert_affine(fwd).reshape(B, 6)).to(n.device)
        check(L.vu_denoise_prepare(ptr(n), ptr(c), ptr(x), ptr(y), ptr(scratch), scratch.numel(), ptr(minv), B, H, W, Cn, im,
                                   self.mean, self.std, stream_ptr(n.device)), "vu_denoise_prepare")
        return {"x": x, "y": y}
