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


def shift_scale_rotate_matrices(B: int, im, shift_limit: float = 0.2, scale_limit: float = 0.2,
                                rotate_limit: float = 20.0, rng: Optional[random.Random] = None) -> np.ndarray:
    """Per-image forward 2x3 matrices of albumentations.ShiftScaleRotate(p=1.0) as
    run_denoising.py:53 configures it: angle ~ U(-rotate_limit, rotate_limit) degrees,
    scale ~ U(1 - scale_limit, 1 + scale_limit), dx, dy ~ U(-shift_limit, shift_limit) (fractions of
    the image size); rotation about (w/2 - 0.5, h/2 - 0.5)."""
    rng = rng or random
    out = np.zeros((B, 2, 3), dtype=np.float64)
    h, w = (im, im) if isinstance(im, int) else im
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5
    for b in range(B):
        angle = rng.uniform(-rotate_limit, rotate_limit)
        scale = rng.uniform(1.0 - scale_limit, 1.0 + scale_limit)
        dx, dy = rng.uniform(-shift_limit, shift_limit), rng.uniform(-shift_limit, shift_limit)
        a = scale * math.cos(math.radians(angle))
        s = scale * math.sin(math.radians(angle))
        out[b] = [[a, s, (1 - a) * cx - s * cy + dx * w], [-s, a, s * cx + (1 - a) * cy + dy * h]]
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



class SegmentationBatchTransform:
    """Device-side scaling / augmentation for SegmentationDataset batches (the reference leaves both
    to the caller's `augments`, dataset.py:32-36): resize to `im_size` (bilinear slice, nearest
    mask) -> ShiftScaleRotate when `train` -> intensity window `window` -> [0,1]; mask ->
    label (1 - ls) + ls / 2.  Call with image (B,H,W) int16 and mask (B,H,W) uint8 (either may be
    None); returns {'x': (B,1,h,w) float32, 'y': ...} on the device (C ABI `vu_seg_prepare`)."""

    def __init__(self, im_size=(128, 128), train: bool = True, window=(-1024.0, 1024.0), ls: float = 0.0,
                 device="cuda", seed: Optional[int] = None):
        self.im_size = (int(im_size), int(im_size)) if isinstance(im_size, int) else (int(im_size[0]), int(im_size[1]))
        self.train, self.window, self.ls, self.device = bool(train), (float(window[0]), float(window[1])), float(ls), device
        self.rng = random.Random(seed)

    def _dev(self, a, dtype) -> Optional[torch.Tensor]:
        if a is None:
            return None
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        assert t.dtype == dtype and t.dim() == 3, f"expected a (B,H,W) {dtype} batch, got {t.dtype} {tuple(t.shape)}"
        return t.to(self.device, non_blocking=True).contiguous()

    def __call__(self, image, mask=None, matrices: Optional[np.ndarray] = None) -> Dict[str, torch.Tensor]:
        i, m = self._dev(image, torch.int16), self._dev(mask, torch.uint8)
        ref = i if i is not None else m
        assert ref is not None, "nothing to transform"
        assert i is None or m is None or i.shape == m.shape, "image and mask shapes differ"
        B, H, W = ref.shape
        oh, ow = self.im_size
        L = lib()
        x = torch.empty(B, 1, oh, ow, dtype=torch.float32, device=ref.device) if i is not None else None
        y = torch.empty(B, 1, oh, ow, dtype=torch.float32, device=ref.device) if m is not None else None
        scratch = torch.empty(max(L.vu_seg_prepare_scratch_bytes(B, oh, ow), 2), dtype=torch.uint8, device=ref.device)
        minv = None
        if self.train:
            fwd = matrices if matrices is not None else shift_scale_rotate_matrices(B, (oh, ow), rng=self.rng)
            assert np.asarray(fwd).shape == (B, 2, 3), "matrices must be (B,2,3)"
            minv = torch.from_numpy(invert_affine(fwd).reshape(B, 6)).to(ref.device)
        check(L.vu_seg_prepare(ptr(i), ptr(m), ptr(x), ptr(y), ptr(scratch), scratch.numel(), ptr(minv), B, H, W, oh, ow,
                               self.window[0], self.window[1], self.ls, stream_ptr(ref.device)), "vu_seg_prepare")
        out = {}
        if x is not None:
            out["x"] = x
        if y is not None:
            out["y"] = y
        return out


def _read_png_bgr(path: str) -> np.ndarray:
    """cv2.imread(path) without OpenCV: decoded 8-bit pixels, HWC, BGR channel order."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


class DenoisingDataset(torch.utils.data.Dataset):
    """dataset.py:44-73, host half only: an item is the DECODED pair {'x': noisy, 'y': clean} as
    (H,W,3) uint8 BGR arrays.  The reference's per-item resize / augment / normalise / CHW steps run
    per batch on the GPU: pass a `DenoisingBatchTransform` as `augments` and iterate with
    `DeviceBatches` (or call `dataset.augments(noisy_batch, clean_batch)` yourself).  `read` replaces
    the decoder (default: PIL in cv2.imread's BGR order)."""

    def __init__(self, img_names, augments=None, clean_folder='/ssid/clean/', noisy_folder='/ssid/noisy/', im_size=224,
                 read=None):
        import os
        self._join = os.path.join
        self.img_names, self.augments = img_names, augments
        self.clean_folder, self.noisy_folder, self.im_size = clean_folder, noisy_folder, im_size
        self.read = read or _read_png_bgr
        if augments is not None and not isinstance(augments, DenoisingBatchTransform):
            raise TypeError("augments must be a DenoisingBatchTransform (the albumentations pipeline of "
                            "run_denoising.py:52-59 runs on the device here) or None")

    def __getitem__(self, idx):
        name = self.img_names[idx]
        return {'x': self.read(self._join(self.noisy_folder, name) + '.png'),
                'y': self.read(self._join(self.clean_folder, name) + '.png')}

    def __len__(self):
        return len(self.img_names)

    def transform(self, items) -> Dict[str, torch.Tensor]:
        """A list of items -> one device batch {'x','y'} (B,3,im,im) float32."""
        t = self.augments or DenoisingBatchTransform(im_size=self.im_size, train=False)
        return _batched(items, lambda xs, ys: t(np.stack(xs), np.stack(ys)))


class SegmentationDataset(torch.utils.data.Dataset):
    """dataset.py:9-41, host half only: an item is {'x': DICOM slice (H,W) int16, 'y': mask plane
    (H,W) uint8 labels} as read (`read_image(path)` / `read_mask(path, index)`; defaults: pydicom
    pixel_array and nibabel get_fdata()[:, :, index], imported lazily).  Scaling / augmentation runs
    per batch on the GPU: `augments` is a `SegmentationBatchTransform` (default: resize to `im_size`,
    label smoothing `ls`, no warp).  `is_test`: items carry no mask."""

    def __init__(self, df, augments=None, is_test=False, data_folder='output', im_size=(128, 128), ls=0.0,
                 read_image=None, read_mask=None):
        self.df, self.augments, self.is_test = df, augments, is_test
        self.data_folder, self.im_size, self.ls = data_folder, im_size, ls
        self.read_image, self.read_mask = read_image or self._dicom, read_mask or self._nifti
        if augments is not None and not isinstance(augments, SegmentationBatchTransform):
            raise TypeError("augments must be a SegmentationBatchTransform or None")

    @staticmethod
    def _dicom(path):
        import pydicom
        a = np.asarray(pydicom.read_file(path).pixel_array)
        # the device pipeline takes int16 slices (vu_seg_prepare): unsigned 16-bit DICOM (PixelRepresentation = 0) values
        # >= 32768 would wrap negative in the cast - refuse them instead of silently windowing wrapped intensities
        if a.dtype != np.int16 and a.size and (int(a.max()) > 32767 or int(a.min()) < -32768):
            raise ValueError(f"{path}: pixel values outside the int16 range ({a.dtype}, max {int(a.max())}): pass a `read_image` "
                             "that applies RescaleSlope / RescaleIntercept (or shifts the range) and returns int16")
        return a.astype(np.int16)

    @staticmethod
    def _nifti(path, index):
        import nibabel as nib
        return np.asarray(nib.load(path).get_fdata()[:, :, index]).astype(np.uint8)

    def __getitem__(self, idx):
        row = self.df.loc[idx]
        item = {'x': self.read_image(row['image'])}
        if not self.is_test:
            item['y'] = self.read_mask(row['mask'], row['mask_index'])
        return item

    def __len__(self):
        return len(self.df)

    def transform(self, items) -> Dict[str, torch.Tensor]:
        t = self.augments or SegmentationBatchTransform(im_size=self.im_size, train=False, ls=self.ls)
        if self.is_test:
            return _batched(items, lambda xs, ys: t(np.stack(xs), None), with_y=False)
        return _batched(items, lambda xs, ys: t(np.stack(xs), np.stack(ys)))


def _batched(items, fn, with_y=True):
    """Items of one source size go through `fn` together; mixed sizes are grouped (order kept)."""
    groups = {}
    for i, it in enumerate(items):
        groups.setdefault(np.asarray(it['x']).shape, []).append(i)
    parts, order = [], []
    for idxs in groups.values():
        parts.append(fn([np.asarray(items[i]['x']) for i in idxs], [np.asarray(items[i]['y']) for i in idxs] if with_y else None))
        order += idxs
    if len(parts) == 1:
        return parts[0]
    inv = torch.as_tensor(np.argsort(order), device=parts[0]['x'].device)
    return {k: torch.cat([p[k] for p in parts]).index_select(0, inv) for k in parts[0]}


class DeviceBatches:
    """The DataLoader of run_denoising.py:60-75 for the device pipeline: host workers decode items
    (`num_workers` DataLoader workers, identity collate), the dataset's batch transform runs on the
    GPU, and what comes out is the {'x','y'} dict `ImageFitter.unpack` takes."""

    def __init__(self, dataset, batch_size=1, shuffle=False, num_workers=0, drop_last=False):
        self.dataset = dataset
        self.loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                                                  drop_last=drop_last, collate_fn=list)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for items in self.loader:
            yield self.dataset.transform(items)
