"""Training harness next to the model: the counterpart of the reference's `ImageFitter`
(vit_unet/torch/dataset.py:76-91, a benatools `TorchFitterBase`) as `run_denoising.py:84-100`
drives it - `ImageFitter(model, loss=, optimizer=, device=, folder=)`, `fit(train, val, n_epochs,
callbacks)`, best / last checkpoints under `folder`, `load(path)`, `.model`.

benatools is not part of the reference tree (unpinned dependency), so only the
behaviour the reference relies on is reproduced: `unpack` -> `model(x)` -> loss -> backward ->
optimizer step per batch, a validation pass per epoch, `best-checkpoint.bin` written when the
monitored loss improves, `last-checkpoint.bin` every epoch, and callbacks that receive a dict
containing `'epoch'`.

When the loss is `MSELoss` (or None) or `functions.DiceLoss()` and the optimizer is `AdamW` (or None)
the batch step is the fused HIP step (`engine.TrainStep`: forward + loss + backward + AdamW without
autograd); any other loss / optimizer runs through the module's autograd path.
"""
from __future__ import annotations

import os
import warnings
from typing import Callable, Dict, Iterable, List, Optional

import torch


class ImageFitter:
    def __init__(self, model, loss=None, optimizer=None, device="cuda", folder="models", lr: float = 1e-4, seed: int = 0):
        self.model = model
        self.loss = loss if loss is not None else torch.nn.MSELoss()
        self.optimizer = optimizer
        self.device = device
        self.folder = folder
        self.epoch = 0
        self.best_metric = float("inf")
        self._lr = lr
        self._seed = seed
        self._fused = None
        self._fused_state = None         # optimizer state restored by load(), applied when the fused step is (re)built
        self._torch_state = None         # torch optimizer state restored by load() before the autograd path made its optimizer
        self._last_hyper = None

    # ---- reference surface -----------------------------------------------------------------------
    def unpack(self, data):
        """dataset.py:78-91: x, y (and optional per-sample weights w) as float tensors on the device."""
        x = data["x"].to(self.device).float()
        y = data["y"].to(self.device).float()
        w = data["w"].to(self.device).float() if "w" in data else None
        return x, y, w

    def _fused_kind(self) -> Optional[str]:
        from .functions import DiceLoss
        if isinstance(self.loss, torch.nn.MSELoss) and getattr(self.loss, "reduction", "mean") == "mean":
            return "mse"
        if isinstance(self.loss, DiceLoss) and self.loss.apply_sigmoid:
            return "dice"
        return None

    def _fused_ok(self) -> bool:
        if self._fused_kind() is None:
            return False
        if self.optimizer is not None and type(self.optimizer) is not torch.optim.AdamW:
            return False
        if self.optimizer is not None and len(self.optimizer.param_groups) != 1:
            return False
        return str(self.device).startswith("cuda")

    def _make_fused(self):
        from .engine import TrainStep
        kw = dict(lr=self._lr)
        if self.optimizer is not None:
            g = self.optimizer.param_groups[0]
            kw = dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])
        ts = TrainStep(self.model, seed=self._seed, loss=self._fused_kind(), **kw)
        if self._fused_state is not None:
            ts.load_state_dict(self._fused_state)
            self._fused_state = None
        if self._torch_state is not None:
            warnings.warn("ImageFitter: the checkpoint holds torch-optimizer state but this run takes the fused HIP step; "
                          "the optimizer moments start from zero")
            self._torch_state = None
        return ts

    def _sync_hyper(self):
        """The fused step reads lr / betas / eps / weight_decay from the user's optimizer param group on EVERY batch, so
        an LR scheduler or a manual param_groups[0]['lr'] edit takes effect as it would with optimizer.step()."""
        if self.optimizer is None:
            return
        g = self.optimizer.param_groups[0]
        cur = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
        if cur != self._last_hyper:
            self._fused.set_hyper(lr=cur[0], betas=(cur[1], cur[2]), eps=cur[3], weight_decay=cur[4])
            self._last_hyper = cur

    def _train_batch(self, x, y, w) -> float:
        if w is None and self._fused_ok():
            if self._fused is None:
                self._fused = self._make_fused()
                self._last_hyper = None
            self._sync_hyper()
            return float(self._fused.step(x, y).item())
        if self.optimizer is None:
            self.optimizer = torch.optim.AdamW(self.model.parameters(), lr=self._lr)
        if self._torch_state is not None:          # a resumed run: the checkpoint's moments, applied now that the optimizer exists
            self.optimizer.load_state_dict(self._torch_state)
            self._torch_state = None
        if self._fused_state is not None:
            warnings.warn("ImageFitter: the checkpoint holds fused-AdamW state but this run takes the autograd path; "
                          "the optimizer moments start from zero")
            self._fused_state = None
        self.optimizer.zero_grad()
        out = self.model(x)
        if w is None:
            loss = self.loss(out, y)
        else:   # per-sample weights: weighted mean of the per-sample criterion
            if isinstance(self.loss, torch.nn.MSELoss):
                per = ((out - y) ** 2).reshape(out.shape[0], -1).mean(dim=1)
            elif hasattr(self.loss, "reduction"):
                crit = type(self.loss)(reduction="none")
                per = crit(out, y).reshape(out.shape[0], -1).mean(dim=1)
            else:
                raise NotImplementedError("per-sample weights need a criterion with reduction='none' (e.g. MSELoss, L1Loss)")
            loss = (per * w.reshape(-1)).sum() / w.sum()
        loss.backward()
        self.optimizer.step()
        return float(loss.item())

    @torch.no_grad()
    def validate(self, loader: Iterable) -> float:
        self.model.eval()
        tot, n = 0.0, 0
        for data in loader:
            x, y, _ = self.unpack(data)
            tot += float(self.loss(self.model(x), y).item()) * x.shape[0]
            n += x.shape[0]
        return tot / max(n, 1)

    def fit(self, train_loader: Iterable, val_loader: Optional[Iterable] = None, n_epochs: int = 1,
            callbacks: Optional[List[Callable[[Dict], None]]] = None, verbose: bool = False) -> List[Dict]:
        os.makedirs(self.folder, exist_ok=True)
        history = []
        for _ in range(n_epochs):
            self.model.train()
            tot, n = 0.0, 0
            for data in train_loader:
                x, y, w = self.unpack(data)
                tot += self._train_batch(x, y, w) * x.shape[0]
                n += x.shape[0]
            log = {"epoch": self.epoch, "train": tot / max(n, 1)}
            monitored = log["train"]
            if val_loader is not None:
                log["val"] = self.validate(val_loader)
                monitored = log["val"]
            self.epoch += 1               # checkpoints record the number of COMPLETED epochs
            self.save(os.path.join(self.folder, "last-checkpoint.bin"))
            if monitored < self.best_metric:
                self.best_metric = monitored
                self.save(os.path.join(self.folder, "best-checkpoint.bin"))
            if verbose:
                print(log)
            for cb in callbacks or []:
                cb(dict(log))
            history.append(log)
        return history

    # ---- checkpoints (reference key names: state_dict of the nn.Module) ----------------------------
    def save(self, path: str):
        """model_state_dict (reference key names) + optimizer_state_dict (the fused AdamW's moments / step in arena
        layout, or the torch optimizer's own state on the autograd path) + epoch / best metric."""
        opt = None
        if self._fused is not None:
            opt = {"kind": "fused_adamw", **self._fused.state_dict()}
        elif self.optimizer is not None:
            opt = {"kind": "torch", "state": self.optimizer.state_dict()}
        torch.save({"model_state_dict": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                    "optimizer_state_dict": opt, "epoch": self.epoch, "best_metric": self.best_metric}, path)

    def load(self, path: str, weights_only: bool = False):
        """Restore a checkpoint.  `weights_only=True` is the reference's use (run_denoising.py:100 reloads the best
        weights for evaluation); otherwise the optimizer state is restored too, so that a resumed run continues the
        interrupted one (same moments, same bias-correction step)."""
        # weights_only: tensors and plain containers only (nothing is unpickled from an untrusted file)
        ck = torch.load(path, map_location="cpu", weights_only=bool(weights_only))
        self.model.load_state_dict(ck["model_state_dict"])     # (the model's post-hook invalidates the bf16 shadow)
        self.epoch = ck.get("epoch", 0)
        self.best_metric = ck.get("best_metric", float("inf"))
        self._fused = None
        self._fused_state = None
        self._torch_state = None
        opt = ck.get("optimizer_state_dict")
        if opt is not None and not weights_only:
            if opt.get("kind") == "fused_adamw":
                self._fused_state = {k: v for k, v in opt.items() if k != "kind"}
            elif opt.get("kind") == "torch":
                if self.optimizer is not None:
                    self.optimizer.load_state_dict(opt["state"])
                else:
                    self._torch_state = opt["state"]       # applied when the autograd path creates its optimizer
        return self
