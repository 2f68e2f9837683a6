"""K-fold denoising run: the counterpart of `run_denoising.py:16-122` without W&B, fire and file
decoding (those stay with the caller) - per fold: build the preset model, `AdamW`, `ImageFitter`,
`fit(train, val, n_epochs, callbacks)`, reload `best-checkpoint.bin`, per-image PSNR of the test
split.  The loaders feed decoded uint8 HWC images through the device-side input pipeline
(`dataset.DenoisingBatchTransform`: train transform on the training split, validation transform on
the test split, run_denoising.py:52-59)."""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import functions as fn
from . import model as models
from .dataset import DenoisingBatchTransform, ImageFitter


def kfold_indices(n: int, folds: int, seed: Optional[int] = None):
    """sklearn.model_selection.KFold(folds, shuffle=True).split: the first n % folds folds get one
    extra sample; yields (train_idx, test_idx) with sorted indices."""
    assert 2 <= folds <= n, f"cannot split {n} samples into {folds} folds"
    perm = np.random.RandomState(seed).permutation(n)
    sizes = np.full(folds, n // folds)
    sizes[: n % folds] += 1
    stop = np.cumsum(sizes)
    for f in range(folds):
        test = np.sort(perm[stop[f] - sizes[f]: stop[f]])
        mask = np.ones(n, bool)
        mask[test] = False
        yield np.nonzero(mask)[0], test


class BatchLoader:
    """Minimal stand-in for the reference's DataLoader over DenoisingDataset: batches of decoded uint8
    images -> {'x','y'} device tensors via the device pipeline.  Re-iterable (one pass per epoch)."""

    def __init__(self, noisy: np.ndarray, clean: np.ndarray, idx: Sequence[int], batch_size: int, transform,
                 shuffle: bool, seed: int = 0):
        self.noisy, self.clean, self.idx, self.bs, self.tf, self.shuffle = noisy, clean, np.asarray(idx), batch_size, transform, shuffle
        self.rs = np.random.RandomState(seed)

    def __len__(self):
        return (len(self.idx) + self.bs - 1) // self.bs

    def __iter__(self):
        order = self.rs.permutation(self.idx) if self.shuffle else self.idx
        for i in range(0, len(order), self.bs):
            sel = order[i:i + self.bs]
            yield self.tf(self.noisy[sel], self.clean[sel])


def run_denoising(noisy: np.ndarray, clean: np.ndarray, n_epochs: int = 5, folds: int = 5, model_string: str = "lite",
                  lr: float = 1e-4, batch_size: int = 8, im_size: int = 224, folder: str = "models", seed: int = 0,
                  dtype=torch.bfloat16, callbacks: Optional[List[Callable[[int, Dict], None]]] = None,
                  verbose: bool = False) -> Dict:
    """noisy / clean: (n,H,W,3) uint8 decoded images (what `cv2.imread` returns for the SIDD pairs).
    Returns {'psnr': [per-fold arrays], 'psnr_mean', 'psnr_std', 'history'} (run_denoising.py:113-119)."""
    assert len(clean) == len(noisy), f"Clean length {len(clean)} is not equal to Noisy length {len(noisy)}"
    results, histories = [], []
    for fold, (train_idx, test_idx) in enumerate(kfold_indices(len(noisy), folds, seed)):
        if verbose:
            print(f"FOLD {fold}: Training on {len(train_idx)} samples and testing on {len(test_idx)} samples")
        train = BatchLoader(noisy, clean, train_idx, batch_size, DenoisingBatchTransform(im_size, train=True, seed=seed + fold),
                            shuffle=True, seed=seed + fold)
        test = BatchLoader(noisy, clean, test_idx, batch_size, DenoisingBatchTransform(im_size, train=False), shuffle=False)
        model = models.get_vit_unet(model_string, dtype=dtype)
        model.to("cuda")
        criterion = torch.nn.MSELoss()
        optimizer = torch.optim.AdamW(model.parameters(), lr=lr)
        fitter = ImageFitter(model, loss=criterion, optimizer=optimizer, device="cuda", folder=folder, seed=seed + fold)
        cbs = [(lambda log, f=fold, cb=cb: cb(f, log)) for cb in (callbacks or [])]
        histories.append(fitter.fit(train, test, n_epochs=n_epochs, callbacks=cbs, verbose=verbose))
        fitter.load(os.path.join(folder, "best-checkpoint.bin"))
        model = fitter.model
        model.eval()
        score = fn.psnr(model, test)
        if verbose:
            print(f"FOLD {fold}: Mean PSNR {np.mean(score)}")
        results.append(score)
    # run_denoising.py:109-112: np.mean(results) / np.std(results) over the list of per-fold score arrays, i.e. over ALL
    # per-image scores of all folds (not the spread of the fold means)
    allv = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1) for r in results])
    return {"psnr": results, "psnr_mean": float(allv.mean()), "psnr_std": float(allv.std()), "history": histories}
