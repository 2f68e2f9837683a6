"""Metrics / losses next to the model (reference: vit_unet/torch/functions.py:7-19, README.md:85-101).

`psnr(model, dataloader)` keeps the reference signature but computes the metric on the device
(10*log10(R^2/MSE) per image, data_range 1.0 = what scikit-image infers for float targets in
[0,1]) and transfers one vector per batch instead of the whole output tensor."""
from __future__ import annotations

import numpy as np
import torch


def psnr_batch(target: torch.Tensor, out: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    B = target.shape[0]
    mse = ((target.double() - out.double()) ** 2).reshape(B, -1).mean(dim=1)
    return 10.0 * torch.log10(data_range ** 2 / mse)


def psnr(model, dataloader, device="cuda"):
    score = []
    with torch.no_grad():
        for batch in dataloader:
            x = batch["x"].to(device).float()
            y = batch["y"].to(device).float()
            score.append(psnr_batch(y, model(x)).cpu().numpy())
    return np.concatenate(score) if score else np.zeros(0)


def dice_loss(input: torch.Tensor, target: torch.Tensor):
    """README.md:91-101."""
    smooth = 1.0
    iflat, tflat = input.reshape(-1), target.reshape(-1)
    intersection = (iflat * tflat).sum()
    return 1 - ((2.0 * intersection + smooth) / (iflat.sum() + tflat.sum() + smooth))
