"""Metrics / losses next to the model (reference: vit_unet/torch/functions.py:7-19, README.md:85-101).

`psnr(model, dataloader)` keeps the reference signature but computes the metric on the device with
the HIP kernels of csrc/vu_metrics.hip (10*log10(R^2/MSE) per image, data_range 1.0 = what
scikit-image infers for float targets in [0,1]) and transfers one vector per batch instead of the
whole output tensor.  `ssim_batch` / `dice_loss` are the two other measures the README names;
`dice_loss` is differentiable (its backward is the HIP gradient kernel).  No CPU fallback."""
from __future__ import annotations

import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


def psnr_batch(target: torch.Tensor, out: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Per-image PSNR of two (B, ...) GPU tensors -> (B,) float32 on the device."""
    assert target.shape == out.shape, f"shape mismatch {tuple(target.shape)} vs {tuple(out.shape)}"
    t, o = _f32(target), _f32(out)
    B = t.shape[0]
    P = t.numel() // max(B, 1)
    L = lib()
    res = torch.empty(B, dtype=torch.float32, device=t.device)
    part = torch.empty(max(1, L.vu_psnr_partials_floats(B)), dtype=torch.float32, device=t.device)
    check(L.vu_psnr(ptr(t), ptr(o), ptr(res), ptr(part), B, P, float(data_range), stream_ptr(t.device)), "vu_psnr")
    return res


def ssim_batch(target: torch.Tensor, out: torch.Tensor, data_range: float = 1.0, win_size: int = 7) -> torch.Tensor:
    """Per-image mean SSIM (channels averaged) of two (B,C,H,W) GPU tensors -> (B,) float32."""
    assert target.shape == out.shape and target.dim() == 4, "ssim_batch takes two (B,C,H,W) tensors"
    t, o = _f32(target), _f32(out)
    B, Cn, H, W = t.shape
    L = lib()
    res = torch.empty(B, dtype=torch.float32, device=t.device)
    part = torch.empty(max(1, L.vu_ssim_partials_floats(B, Cn, H, W, win_size)), dtype=torch.float32, device=t.device)
    check(L.vu_ssim(ptr(t), ptr(o), ptr(res), ptr(part), B, Cn, H, W, int(win_size), float(data_range),
                    stream_ptr(t.device)), "vu_ssim")
    return res


def psnr(model, dataloader, device="cuda"):
    score = []
    with torch.no_grad():
        for batch in dataloader:
            x = batch["x"].to(device).float()
            y = batch["y"].to(device).float()
            score.append(psnr_batch(y, model(x)).cpu().numpy())
    return np.concatenate(score) if score else np.zeros(0)


def ssim(model, dataloader, device="cuda"):
    """Same calling convention as `psnr` (the README lists SSIM beside PSNR)."""
    score = []
    with torch.no_grad():
        for batch in dataloader:
            x = batch["x"].to(device).float()
            y = batch["y"].to(device).float()
            score.append(ssim_batch(y, model(x)).cpu().numpy())
    return np.concatenate(score) if score else np.zeros(0)


class _Dice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target, apply_sigmoid):
        x, t = inp.detach().float().contiguous(), _f32(target)
        L = lib()
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        part = torch.empty(L.vu_dice_partials_floats(), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        check(L.vu_dice_loss(ptr(x), ptr(t), ptr(dx), ptr(loss), ptr(part), x.numel(), int(apply_sigmoid), 1.0,
                             stream_ptr(x.device)), "vu_dice_loss")
        ctx.save_for_backward(dx)
        ctx.in_dtype, ctx.in_shape = inp.dtype, inp.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return (dx * g).reshape(ctx.in_shape).to(ctx.in_dtype), None, None


def dice_loss(input: torch.Tensor, target: torch.Tensor, apply_sigmoid: bool = False):
    """README.md:91-101 (smooth = 1, flattened).  `apply_sigmoid=True` is the segmentation head of
    BASELINE config 5: `input` are the model's logits."""
    assert input.numel() == target.numel(), "dice_loss: input and target sizes differ"
    return _Dice.apply(input, target, apply_sigmoid)


class DiceLoss(torch.nn.Module):
    """Criterion object for `ImageFitter(model, loss=DiceLoss())`: Dice on sigmoid(model output) (the
    segmentation configuration, BASELINE config 5).  With `apply_sigmoid=True` the fitter runs the
    fused HIP step (`TrainStep(loss="dice")`)."""

    def __init__(self, apply_sigmoid: bool = True):
        super().__init__()
        self.apply_sigmoid = bool(apply_sigmoid)

    def forward(self, input, target):
        return dice_loss(input, target, apply_sigmoid=self.apply_sigmoid)
