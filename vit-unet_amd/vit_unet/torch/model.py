"""MI355X-native ViT-UNet: the nn.Module surface of the reference model on top of HIP kernels.

Mirrors `/root/reference/vit_unet/torch/model.py` for the hot path only: the same free functions
(`patch`, `unflatten`, `unpatch`, `downsampling`, `upsampling`), the same classes, attribute and
state_dict names (`PE, Encoders, BottleNeck, Decoders, SkipConnections, conv2d`, inner `ReAttn,
LN1, LN2, FeedForward.net.{0,3}, reatten_matrix, var_norm, qconv2d, kconv2d, vconv2d, proj,
position_embedding`), `get_vit_unet('lite'|'base'|'large')` (model.py:438-486) and the README's
`ViT_UNet(...)` constructor (README.md:18-31).  Every tensor operation runs in
`libvitunet_amd.so` (csrc/): PyTorch only owns memory, the current stream and the autograd edge.

Design: all parameters are views of ONE flat fp32 arena (gradients likewise),
laid out by the C side (`vu_model_param_table`); a whole-model forward / backward is one C call
each (`vu_model_forward` / `vu_model_backward`) that enqueues every kernel on the current stream.
`torch.nn.Conv2d / Linear / LayerNorm / BatchNorm2d / Embedding` objects are used purely as
parameter containers (default initialisers and state_dict keys identical to the reference); their
own `forward` is never called.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

nn = torch.nn


# ---------------------------------------------------------------------------------------------
# re-tiling helpers (model.py:8-53) - permutations done by the vu_retile kernel
# ---------------------------------------------------------------------------------------------
def _retile(x: torch.Tensor, C_: int, im: int, s_in: int, s_out: int) -> torch.Tensor:
    if x.dtype not in _lib.DTYPE_CODE:
        raise TypeError("retile: float32 or bfloat16 tensors only")
    xin = x.contiguous()
    B = xin.shape[0]
    e = im // s_out
    out = torch.empty(B, e * e, C_ * s_out * s_out, dtype=x.dtype, device=x.device)
    check(lib().vu_retile(_lib.DTYPE_CODE[x.dtype], 0, 0, ptr(xin), ptr(out), None, B, C_, im, s_in, s_out,
                          stream_ptr(x.device)), "vu_retile")
    return out


class _RetileFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C_, im, s_in, s_out):
        ctx.args = (C_, im, s_in, s_out)
        return _retile(x, C_, im, s_in, s_out)

    @staticmethod
    def backward(ctx, g):
        C_, im, s_in, s_out = ctx.args
        return _retile(g, C_, im, s_out, s_in), None, None, None, None


def patch(X: torch.Tensor, patch_size: int):
    """(B,C,H,W) -> (B, n_patches, C, p, p)   (model.py:8-18)"""
    if X.dim() == 5:
        X = torch.squeeze(X, dim=1)
    B, C_, h, w = X.shape
    assert h % patch_size == 0, "Patch size must divide images height"
    assert w % patch_size == 0, "Patch size must divide images width"
    assert h == w, "the HIP re-tiling kernel handles square images"
    t = _RetileFn.apply(X.reshape(B, 1, C_ * h * w), C_, h, h, patch_size)
    return t.reshape(B, -1, C_, patch_size, patch_size)


def unflatten(flattened: torch.Tensor, num_channels: int):
    """(B,N,D) -> (B,N,C,s,s)   (model.py:20-24); a view."""
    bs, n, p = flattened.size()
    s = int(np.sqrt(p // num_channels))
    return torch.reshape(flattened, (bs, n, num_channels, s, s))


def unpatch(x: torch.Tensor, num_channels: int):
    """inverse of patch -> (B,1,C,H,W)   (model.py:26-35)"""
    if x.dim() < 5:
        x = unflatten(x, num_channels)
    B, N, ch, h, w = x.size()
    assert ch == num_channels, "Num. channels must agree"
    e = int(np.sqrt(N))
    im = e * h
    t = _RetileFn.apply(x.reshape(B, N, ch * h * w), ch, im, h, im)
    return t.reshape(B, 1, ch, im, im)


def _resample(encoded_patches: torch.Tensor, num_channels: int, factor_num: int, factor_den: int):
    B, N, D = encoded_patches.shape
    s = int(np.sqrt(D / num_channels))
    im = int(np.sqrt(N)) * s
    return _RetileFn.apply(encoded_patches, num_channels, im, s, s * factor_num // factor_den)


def downsampling(encoded_patches: torch.Tensor, num_channels: int):
    """same latent image, patch size halved: (B,N,D) -> (B,4N,D/4)   (model.py:39-45)"""
    return _resample(encoded_patches, num_channels, 1, 2)


def upsampling(encoded_patches: torch.Tensor, num_channels: int):
    """same latent image, patch size doubled: (B,N,D) -> (B,N/4,4D)   (model.py:47-53)"""
    return _resample(encoded_patches, num_channels, 2, 1)


# ---------------------------------------------------------------------------------------------
# per-op autograd functions for the stand-alone sub-modules
# ---------------------------------------------------------------------------------------------
def _attn_param_struct(m, dtype, need_shadow=True):
    """m: a ReAttention / SkipConnection module.  Returns (vu_attn_params, keepalive list)."""
    keep = []

    def f32(t):
        t = t.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
        keep.append(t)
        return t.data_ptr()
    pw = m.proj.weight.detach()
    if dtype == torch.bfloat16:
        pw = pw.to(torch.bfloat16).contiguous()
    else:
        pw = pw.float().contiguous()
    keep.append(pw)
    def k3(w):      # a 1 x 1 q/k/v kernel (the notebook variant, ViT_UNet.ipynb ReAttention) is the centre tap of a 3 x 3 one
        return w if w.shape[-1] == 3 else torch.nn.functional.pad(w.detach(), (1, 1, 1, 1))
    p = _lib.vu_attn_params(f32(m.reatten_matrix.weight), f32(m.reatten_matrix.bias), f32(m.var_norm.weight),
                            f32(m.var_norm.bias), f32(k3(m.qconv2d.weight)), f32(k3(m.kconv2d.weight)), f32(k3(m.vconv2d.weight)),
                            pw.data_ptr(), f32(m.proj.bias), m.var_norm.running_mean.data_ptr(),
                            m.var_norm.running_var.data_ptr(), _lib.operand_code(getattr(m, "attn_operands", "storage")))
    return p, keep


_ATTN_PARAM_NAMES = ["reatten_matrix.weight", "reatten_matrix.bias", "var_norm.weight", "var_norm.bias",
                     "qconv2d.weight", "kconv2d.weight", "vconv2d.weight", "proj.weight", "proj.bias"]


class _AttnFn(torch.autograd.Function):
    """ReAttention / SkipConnection through vu_attn_forward / vu_attn_backward."""

    @staticmethod
    def forward(ctx, xq, xkv, module, training, seed, stream_id, want_map, *params):
        dt = xq.dtype
        code = _lib.DTYPE_CODE[dt]
        B, N, D = xq.shape
        H, Cn = module.num_heads, module.num_channels
        xq_c, xkv_c = xq.contiguous(), xkv.contiguous()
        L = lib()
        nbytes = L.vu_attn_workspace_bytes(code, B, N, D, H)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xq.device)
        y = torch.empty_like(xq_c)
        amap = torch.empty(B, H, N, N, dtype=dt, device=xq.device) if want_map else None
        prm, keep = _attn_param_struct(module, dt)
        check(L.vu_attn_forward(code, C.byref(prm), ptr(xq_c), ptr(xkv_c), ptr(y), ptr(amap), ptr(ws), nbytes,
                                B, N, D, H, Cn, float(module.attn_drop.p), float(module.proj_drop.p),
                                1 if training else 0, seed, stream_id, stream_ptr(xq.device)), "vu_attn_forward")
        if training:
            module.var_norm.num_batches_tracked += 1
        ctx.saved = (xq_c, xkv_c, ws, module, training, seed, stream_id, xq is xkv or xq.data_ptr() == xkv.data_ptr())
        ctx.mark_non_differentiable(*([amap] if want_map else []))
        return (y, amap) if want_map else y

    @staticmethod
    def backward(ctx, dy, *unused):
        xq, xkv, ws, module, training, seed, stream_id, same = ctx.saved
        dt = xq.dtype
        code = _lib.DTYPE_CODE[dt]
        B, N, D = xq.shape
        H, Cn = module.num_heads, module.num_channels
        L = lib()
        prm, keep = _attn_param_struct(module, dt)
        shapes = [(H, H, 1, 1), (H,), (H,), (H,), (Cn, Cn, 3, 3), (Cn, Cn, 3, 3), (Cn, Cn, 3, 3), (D, D), (D,)]
        grads = [torch.zeros(s, dtype=torch.float32, device=xq.device) for s in shapes]
        gs = _lib.vu_attn_grads(*[g.data_ptr() for g in grads])
        dxq = torch.empty_like(xq)
        dxkv = None if same else torch.empty_like(xkv)
        check(L.vu_attn_backward(code, C.byref(prm), C.byref(gs), ptr(xq), ptr(xkv), ptr(dy.contiguous()), ptr(dxq),
                                 ptr(dxkv), ptr(ws), ws.numel(), B, N, D, H, Cn, float(module.attn_drop.p),
                                 float(module.proj_drop.p), 1 if training else 0, seed, stream_id,
                                 stream_ptr(xq.device)), "vu_attn_backward")
        if same:
            gq, gkv = dxq, None
        else:
            gq, gkv = dxq, dxkv
        if module.qconv2d.weight.shape[-1] == 1:      # 1 x 1 kernels: the gradient of the centre tap
            for i in (4, 5, 6):
                grads[i] = grads[i][:, :, 1:2, 1:2].contiguous()
        return (gq, gkv, None, None, None, None, None, *grads)


class _AddLayerNormFn(torch.autograd.Function):
    """LayerNorm((N,D))(a + x) through vu_add_layernorm_fwd / vu_layernorm_bwd."""

    @staticmethod
    def forward(ctx, a, x, w, b):
        dt = a.dtype
        code = _lib.DTYPE_CODE[dt]
        B = a.shape[0]
        P = a[0].numel()
        L = lib()
        a_c = a.contiguous()
        x_c = x.contiguous() if x is not None else None
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        ws = torch.empty(L.vu_layernorm_workspace_floats(B, P), dtype=torch.float32, device=a.device)
        z, y = torch.empty_like(a_c), torch.empty_like(a_c)
        stats = torch.empty(B, 2, dtype=torch.float32, device=a.device)
        check(L.vu_add_layernorm_fwd(code, ptr(a_c), ptr(x_c), ptr(z), ptr(wf), ptr(bf), ptr(y), ptr(ws), ptr(stats),
                                     B, P, stream_ptr(a.device)), "vu_add_layernorm_fwd")
        ctx.saved = (z, wf, stats, ws, x is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, wf, stats, ws, has_x = ctx.saved
        code = _lib.DTYPE_CODE[z.dtype]
        B = z.shape[0]
        P = z[0].numel()
        dw, db = torch.zeros_like(wf), torch.zeros_like(wf)
        dz = torch.empty_like(z)
        check(lib().vu_layernorm_bwd(code, ptr(dy.contiguous()), ptr(z), ptr(wf), ptr(stats), ptr(dw), ptr(db), ptr(ws),
                                     ptr(dz), B, P, stream_ptr(z.device)), "vu_layernorm_bwd")
        return dz, (dz if has_x else None), dw, db


def _next_seed() -> int:
    """Dropout seed for stand-alone module calls, drawn from torch's CPU generator so that
    torch.manual_seed() makes runs reproducible."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


# ---------------------------------------------------------------------------------------------
# modules
# ---------------------------------------------------------------------------------------------
class PatchEncoder(nn.Module):
    """tokens(X) + positional embedding   (model.py:57-91; the conv declared at :79 is never used
    by the reference forward and is not created here - spec decision D1)."""

    def __init__(self, img_size: int, patch_size: int, num_channels: int, projection_dim: Optional[int] = None):
        super().__init__()
        self.img_size, self.patch_size, self.num_channels = img_size, patch_size, num_channels
        self.projection_dim = projection_dim if projection_dim is not None else num_channels * patch_size ** 2
        self.num_patches = (img_size // patch_size) ** 2
        self.register_buffer("positions", torch.arange(self.num_patches), persistent=False)   # D6
        self.position_embedding = nn.Embedding(self.num_patches, self.projection_dim)

    def forward(self, X):
        B = X.shape[0]
        Xc = X.float().contiguous()
        out = torch.empty(B, self.num_patches, self.projection_dim, dtype=torch.float32, device=X.device)
        pos = self.position_embedding.weight.detach().float().contiguous()
        check(lib().vu_retile(0, 1, 1, ptr(Xc), ptr(out), ptr(pos), B, self.num_channels, self.img_size,
                              self.img_size, self.patch_size, stream_ptr(X.device)), "vu_retile")
        return out


class FeedForward(nn.Module):
    """Linear -> GELU -> Dropout -> Linear -> Dropout   (model.py:95-110) on vu_ff_forward / vu_ff_backward: two MFMA
    GEMMs with the bias + exact-erf GELU (+ dropout) and bias (+ dropout) epilogues fused; the backward multiplies by
    GELU' in the epilogue of dh = dy W2.  Inside HViT_UNet the model executor runs the same two launches."""

    def __init__(self, projection_dim: int, hidden_dim: int, dropout: float):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(projection_dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, projection_dim), nn.Dropout(dropout))

    def forward(self, x, seed=None, stream_id=0):
        p = float(self.net[2].p)
        seed = _next_seed() if (seed is None and self.training and p > 0) else (seed or 0)
        return _FeedForwardFn.apply(x, self.net[0].weight, self.net[0].bias, self.net[3].weight, self.net[3].bias,
                                    p, self.training, seed, stream_id)


class ReAttention(nn.Module):
    """model.py:113-164.  apply_transform=False (never used by the model) is not supported."""

    def __init__(self, dim, num_channels=3, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.,
                 proj_drop=0., apply_transform=True, transform_scale=False, qkv_kernel=3):
        super().__init__()
        assert apply_transform and not qkv_bias and qk_scale is None and not transform_scale, \
            "HIP path implements the configuration the model uses (model.py:187-192)"
        assert qkv_kernel in (1, 3), "q/k/v kernels: 3 x 3 (model.py:137-139) or 1 x 1 (the notebook variant)"
        self.num_heads, self.num_channels = num_heads, num_channels
        self.apply_transform = True
        self.scale = (dim // num_heads) ** -0.5
        self.reatten_matrix = nn.Conv2d(num_heads, num_heads, 1, 1)
        self.var_norm = nn.BatchNorm2d(num_heads)
        self.qconv2d = nn.Conv2d(num_channels, num_channels, qkv_kernel, padding="same", bias=False)
        self.kconv2d = nn.Conv2d(num_channels, num_channels, qkv_kernel, padding="same", bias=False)
        self.vconv2d = nn.Conv2d(num_channels, num_channels, qkv_kernel, padding="same", bias=False)
        self.reatten_scale = 1.0
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.attn_operands = "storage"     # or "e4m3": q, k, v rounded to OCP e4m3 before the products (vu_round_e4m3)

    def _params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in _ATTN_PARAM_NAMES]

    def forward(self, x, atten=None, seed=None, stream_id=0):
        seed = _next_seed() if (seed is None and self.training) else (seed or 0)
        y, amap = _AttnFn.apply(x, x, self, self.training, seed, stream_id, True, *self._params())
        return y, amap


class ReAttentionTransformerEncoder(nn.Module):
    """post-norm block LN1(attn(x)+x), LN2(FF(x)+x)   (model.py:167-207)."""

    def __init__(self, num_patches, num_channels, projection_dim, hidden_dim, num_heads, attn_drop, proj_drop,
                 linear_drop):
        super().__init__()
        self.num_patches, self.num_channels, self.projection_dim = num_patches, num_channels, projection_dim
        self.hidden_dim, self.num_heads = hidden_dim, num_heads
        self.attn_drop, self.proj_drop, self.linear_drop = attn_drop, proj_drop, linear_drop
        self.ReAttn = ReAttention(projection_dim, num_channels=num_channels, num_heads=num_heads,
                                  attn_drop=attn_drop, proj_drop=proj_drop)
        self.LN1 = nn.LayerNorm(normalized_shape=(num_patches, projection_dim))
        self.LN2 = nn.LayerNorm(normalized_shape=(num_patches, projection_dim))
        self.FeedForward = FeedForward(projection_dim, hidden_dim, linear_drop)

    def forward(self, encoded_patches, seed=None, stream_id=0):
        # stand-alone use: a one-block model executor call would need a flat arena, so the block is
        # composed from the per-op HIP entry points (attention, residual+LayerNorm) and vu_gemm.
        x = encoded_patches
        seed = _next_seed() if (seed is None and self.training) else (seed or 0)
        a = _AttnFn.apply(x, x, self.ReAttn, self.training, seed, stream_id, False, *self.ReAttn._params())
        x1 = _AddLayerNormFn.apply(a, x, self.LN1.weight, self.LN1.bias)
        f = self.FeedForward(x1, seed=seed, stream_id=stream_id)
        return _AddLayerNormFn.apply(f, x1, self.LN2.weight, self.LN2.bias)


class SkipConnection(nn.Module):
    """cross re-attention merge: q <- encoder skip, k,v <- decoder   (model.py:211-259)."""

    def __init__(self, dim, num_channels=3, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.,
                 transform_scale=False, qkv_kernel=3):
        super().__init__()
        assert not qkv_bias and not transform_scale
        assert qkv_kernel in (1, 3)
        self.num_heads, self.num_channels = num_heads, num_channels
        self.scale = (dim // num_heads) ** -0.5
        self.reatten_matrix = nn.Conv2d(num_heads, num_heads, 1, 1)
        self.var_norm = nn.BatchNorm2d(num_heads)
        self.qconv2d = nn.Conv2d(num_channels, num_channels, qkv_kernel, padding="same", bias=False)
        self.kconv2d = nn.Conv2d(num_channels, num_channels, qkv_kernel, padding="same", bias=False)
        self.vconv2d = nn.Conv2d(num_channels, num_channels, qkv_kernel, padding="same", bias=False)
        self.reatten_scale = 1.0
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.attn_operands = "storage"     # or "e4m3": q, k, v rounded to OCP e4m3 before the products (vu_round_e4m3)

    def _params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in _ATTN_PARAM_NAMES]

    def forward(self, q, k, v, seed=None, stream_id=0):
        assert q.shape == k.shape
        assert k.shape == v.shape
        assert k is v or k.data_ptr() == v.data_ptr(), "the model always passes k is v (model.py:418)"
        seed = _next_seed() if (seed is None and self.training) else (seed or 0)
        return _AttnFn.apply(q, k, self, self.training, seed, stream_id, False, *self._params())


class _FeedForwardFn(torch.autograd.Function):
    """FeedForward through vu_ff_forward / vu_ff_backward (no torch arithmetic)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p, training, seed, stream_id):
        dt = x.dtype
        code = _lib.DTYPE_CODE[dt]
        B, N, D = x.shape
        hid = w1.shape[0]
        xc = x.contiguous()
        w1c, w2c = w1.detach().to(dt).contiguous(), w2.detach().to(dt).contiguous()
        b1c, b2c = b1.detach().float().contiguous(), b2.detach().float().contiguous()
        hpre = torch.empty(B * N, hid, dtype=dt, device=x.device)
        hact = torch.empty_like(hpre)
        y = torch.empty_like(xc)
        check(lib().vu_ff_forward(code, ptr(xc), ptr(w1c), ptr(b1c), ptr(w2c), ptr(b2c), ptr(hpre), ptr(hact), ptr(y),
                                  B * N, D, hid, p, 1 if training else 0, seed, stream_id, stream_ptr(x.device)),
              "vu_ff_forward")
        ctx.saved = (xc, w1c, w2c, hpre, hact, (B, N, D, hid), p, training, seed, stream_id)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, w1c, w2c, hpre, hact, (B, N, D, hid), p, training, seed, stream_id = ctx.saved
        dt = xc.dtype
        code = _lib.DTYPE_CODE[dt]
        L = lib()
        dev = xc.device
        dx = torch.empty_like(xc)
        dw1, db1 = torch.zeros(hid, D, device=dev), torch.zeros(hid, device=dev)
        dw2, db2 = torch.zeros(D, hid, device=dev), torch.zeros(D, device=dev)
        scratch = torch.empty(L.vu_ff_scratch_bytes(code, B * N, D, hid), dtype=torch.uint8, device=dev)
        check(L.vu_ff_backward(code, ptr(xc), ptr(w1c), ptr(w2c), ptr(hpre), ptr(hact), ptr(dy.contiguous().to(dt)),
                               ptr(dx), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(scratch), B * N, D, hid, p,
                               1 if training else 0, seed, stream_id, stream_ptr(dev)), "vu_ff_backward")
        return dx, dw1, db1, dw2, db2, None, None, None, None


# ---------------------------------------------------------------------------------------------
# whole model
# ---------------------------------------------------------------------------------------------
class _ModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, model, training, seed):
        y = model._run_forward(x, training, seed)
        ctx.model, ctx.training, ctx.seed, ctx.gen = model, training, seed, model._gen
        ctx.need_dx = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        m = ctx.model
        if ctx.gen != m._gen:
            raise RuntimeError("HViT_UNet: the activations of this forward were overwritten by a later forward "
                               "(one workspace per model); call backward before the next forward")
        dx = m._run_backward(dy, ctx.training, ctx.seed, ctx.need_dx)
        return dx, None, None, None, None


class HViT_UNet(nn.Module):
    """model.py:263-435.  Extra keyword `dtype` (README ctor surface): torch.float32 runs the
    fp32 kernels (parity mode), torch.bfloat16 runs bf16 storage / fp32 accumulate with fp32
    master weights.  Inputs and outputs are float32 (B,C,im,im) like the reference."""

    def __init__(self, depth: int, depth_te: int, size_bottleneck: int, preprocessing: str, im_size: int,
                 patch_size: int, num_channels: int, hidden_dim: int, num_heads: int, attn_drop: float,
                 proj_drop: float, linear_drop: float, verbose: bool = False, dtype=torch.float32,
                 attn_operands: str = "storage"):
        super().__init__()
        assert patch_size % (2 ** depth) == 0, "Depth must be adjusted, final patch size is incompatible."
        assert patch_size // (2 ** depth) >= 4, "Depth must be adjusted, final patch size is too small (lower than 4)."
        assert im_size % patch_size == 0, "Patch size is not compatible with image size."
        if preprocessing == "fourier":
            raise NotImplementedError("preprocessing='fourier' (model.py:429-430 returns ifft2 of the INPUT) "
                                      "is a reference bug and is not reproduced (spec decision D5)")
        if preprocessing not in ("conv", "none"):
            raise ValueError(f"unknown preprocessing {preprocessing!r}")
        self.depth, self.depth_te, self.size_bottleneck = depth, depth_te, size_bottleneck
        self.preprocessing, self.im_size, self.patch_size = preprocessing, im_size, patch_size
        self.num_patches = (im_size // patch_size) ** 2
        self.num_channels = num_channels
        self.projection_dim = num_channels * patch_size ** 2
        self.hidden_dim, self.num_heads = hidden_dim, num_heads
        self.attn_drop, self.proj_drop, self.linear_drop = attn_drop, proj_drop, linear_drop
        self.verbose = verbose
        self.compute_dtype = dtype
        self._cfg = _lib.make_config(depth, depth_te, size_bottleneck, preprocessing, im_size, patch_size,
                                     num_channels, hidden_dim, num_heads, attn_drop, proj_drop, linear_drop, dtype,
                                     attn_operands)
        self.attn_operands = attn_operands
        if verbose:                                                       # model.py:301-307
            print("Architecture information:")
            for i in range(depth + 1):
                print(f"Level {i}:")
                print("\tPatch size:", patch_size // (2 ** i))
                print("\tNum. patches:", self.num_patches * (4 ** i))
                print("\tProjection size:", self.projection_dim // (4 ** i))
                print("\tHidden dim. size:", hidden_dim // (2 ** i))

        def te(level):
            return ReAttentionTransformerEncoder(self.num_patches * 4 ** level, num_channels,
                                                 self.projection_dim // 4 ** level, hidden_dim // 2 ** level,
                                                 num_heads, attn_drop, proj_drop, linear_drop)
        self.PE = PatchEncoder(im_size, patch_size, num_channels)
        self.Encoders = nn.ModuleList([te(l) for l in range(depth) for _ in range(depth_te)])      # :310-325
        self.BottleNeck = nn.ModuleList([te(depth) for _ in range(size_bottleneck)])                 # :326-340
        self.Decoders = nn.ModuleList([te(depth - l) for l in range(depth) for _ in range(depth_te)])  # :341-358
        self.SkipConnections = nn.ModuleList([
            SkipConnection(dim=self.projection_dim // 4 ** (depth - l - 1), num_channels=num_channels,
                           num_heads=num_heads, attn_drop=attn_drop, proj_drop=proj_drop)
            for l in range(depth)])                                                                  # :359-366
        if preprocessing == "conv":
            self.conv2d = nn.Conv2d(num_channels, num_channels, 3, padding="same")                   # :369-370
        for m in self.modules():
            if isinstance(m, (ReAttention, SkipConnection)):
                m.attn_operands = attn_operands
        # flat-arena state (built lazily: needs the HIP library only once tensors reach the GPU)
        self._arena = self._garena = self._shadow = self._bn = self._ws = None
        self._table = None
        self._gen = 0
        self._shadow_clean = False
        self._shadow_version = None      # weight-version stamp the bf16 shadow was cast from / kept in sync at
        self._nbt_pending = 0            # train-mode forwards not yet added to the num_batches_tracked buffers
        self._step_seed = None
        if os.path.exists(_lib.LIB_PATH):        # reject what the HIP path cannot run at construction, not at first use
            check(lib().vu_model_validate(C.byref(self._cfg)), "vu_model_validate")
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._state_loaded())
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module._flush_bn_counters())

    def _weights_changed(self):
        """Weights were written from outside the fused AdamW (load_state_dict, a torch optimizer, p.mul_ ...): the
        bf16 shadow must be re-cast before the next forward."""
        self._shadow_clean = False

    def _state_loaded(self):
        """load_state_dict post-hook: the loaded num_batches_tracked are the truth - train-mode forwards counted before
        the load must not be added on top of them."""
        self._nbt_pending = 0
        self._weights_changed()

    def _flush_bn_counters(self):
        if self._nbt_pending:
            with torch.no_grad():
                for b in self._bn_modules():
                    b.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0

    def _weight_version(self) -> int:
        """Sum of the autograd version counters of all parameters: any in-place write through torch (optimizer step,
        load_state_dict, p.mul_) changes it; the fused AdamW (a C call) does not and re-stamps the shadow itself."""
        return sum(p._version for p in self._params_list)

    # ---- flat arena ------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._arena = None          # parameters were re-materialised one by one: re-flatten lazily
        return out

    def _bn_modules(self) -> List[nn.BatchNorm2d]:
        mods = [b.ReAttn.var_norm for b in self.Encoders] + [b.ReAttn.var_norm for b in self.BottleNeck]
        mods += [b.ReAttn.var_norm for b in self.Decoders] + [s.var_norm for s in self.SkipConnections]
        return mods

    def _flatten(self):
        """Move every parameter into one fp32 arena laid out by the C side (and gradients into a
        second one) and re-point `.data` / `.grad` at views of it."""
        params = list(self.named_parameters())
        dev = params[0][1].device
        if dev.type != "cuda":
            raise _lib.VuError("HViT_UNet runs on the MI355X only: move the model to 'cuda' (no CPU fallback)")
        for _, p in params:
            if p.dtype != torch.float32:
                raise _lib.VuError("master parameters must stay float32 (use dtype=torch.bfloat16 in the ctor)")
        table = _lib.param_table(self._cfg)
        assert [t[0] for t in table] == [n for n, _ in params], "parameter order differs from the C table"
        total = lib().vu_model_param_elems(C.byref(self._cfg))
        arena = torch.zeros(total, dtype=torch.float32, device=dev)
        garena = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for (name, off, shape, _), (_, p) in zip(table, params):
                assert tuple(p.shape) == shape, (name, tuple(p.shape), shape)
                n = p.numel()
                arena[off:off + n].copy_(p.detach().reshape(-1))
                if p.grad is not None:
                    garena[off:off + n].copy_(p.grad.reshape(-1))
                p.data = arena[off:off + n].view(shape)
                p.grad = garena[off:off + n].view(shape)
        bns = self._bn_modules()
        H = self.num_heads
        bn = torch.zeros(len(bns), 2, H, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for i, b in enumerate(bns):
                bn[i, 0].copy_(b.running_mean)
                bn[i, 1].copy_(b.running_var)
                b.running_mean.data = bn[i, 0]
                b.running_var.data = bn[i, 1]
        self._arena, self._garena, self._bn, self._table = arena, garena, bn, table
        self._anchor = arena.detach().requires_grad_(True)
        self._shadow = torch.empty(total, dtype=torch.bfloat16, device=dev) if self.compute_dtype == torch.bfloat16 else None
        self._shadow_clean = False
        self._ws = None
        self._params_list = [p for _, p in params]
        self._grad_views = [p.grad for _, p in params]

    def _ensure_flat(self):
        if self._arena is None:
            self._flatten()
            return
        # load_state_dict(assign=True) or external code may have replaced Parameter objects or their storage: every
        # registered parameter must still be the cached object and still point at its slot of the arena
        base = self._arena.data_ptr()
        cur = list(self.parameters())
        if len(cur) != len(self._params_list) or any(
                (p is not q) or p.data_ptr() != base + 4 * t[1] for p, q, t in zip(cur, self._params_list, self._table)):
            self._flatten()

    def _link_grads(self):
        """`.grad` of every parameter is a view of the gradient arena.  If an optimizer set grads to
        None (zero_grad(set_to_none=True)) start from a zeroed arena and re-link."""
        stale = False
        for p, gv in zip(self._params_list, self._grad_views):
            if p.grad is not gv:
                stale = True
                break
        if stale:
            with torch.no_grad():
                for p, gv in zip(self._params_list, self._grad_views):
                    if p.grad is None:
                        gv.zero_()
                    elif p.grad is not gv:
                        gv.copy_(p.grad)
                    p.grad = gv

    def _workspace(self, B: int, training: bool = True) -> torch.Tensor:
        """The caller-owned workspace of vu_model_forward / _backward.  An eval forward is sized without the probability caches of
        the recompute attention (vu_model_workspace_bytes_ex(training=0)); a buffer that is already large enough is kept either way.
        If the training size cannot be allocated, the caches are switched off for the process (vu_set_flash_pcache(0): the sweeps
        recompute, bit-identical results) and the smaller layout is tried before the error is raised."""
        L = lib()
        need = L.vu_model_workspace_bytes_ex(C.byref(self._cfg), B, 1 if training else 0)
        if need == 0:
            check(L.vu_model_validate(C.byref(self._cfg)), "vu_model_validate")
        if self._ws is None or self._ws.numel() < need or self._ws.device != self._arena.device:
            self._ws = None
            try:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self._arena.device)
            except torch.OutOfMemoryError:
                if not training or L.vu_model_pcache_bytes(C.byref(self._cfg), B) == 0:
                    raise
                import warnings
                warnings.warn(f"vit_unet: {need / 2 ** 30:.1f} GiB of workspace do not fit; the probability cache of the recompute "
                              f"attention is switched off for this process (VU_FLASH_PCACHE=0)")
                check(L.vu_set_flash_pcache(0), "vu_set_flash_pcache")
                need = L.vu_model_workspace_bytes_ex(C.byref(self._cfg), B, 1)
                self._ws = torch.empty(need, dtype=torch.uint8, device=self._arena.device)
        return self._ws

    def workspace_report(self, B: int) -> dict:
        """Bytes of the training workspace at B images and how much of it is probability cache (bench.py prints it)."""
        L = lib()
        return {"train_bytes": int(L.vu_model_workspace_bytes_ex(C.byref(self._cfg), B, 1)),
                "eval_bytes": int(L.vu_model_workspace_bytes_ex(C.byref(self._cfg), B, 0)),
                "pcache_bytes": int(L.vu_model_pcache_bytes(C.byref(self._cfg), B))}

    def refresh_shadow(self):
        """bf16 copy of the weights for the GEMMs.  A TrainStep keeps it in sync from inside AdamW and vouches for it
        (`_shadow_clean`); the vouching only holds while no torch-side write touched a parameter since (`_weight_version`)."""
        if self._shadow is None:
            return
        ver = self._weight_version()
        if self._shadow_clean and ver == self._shadow_version:
            return
        check(lib().vu_cast_bf16(ptr(self._arena), ptr(self._shadow), self._arena.numel(),
                                 stream_ptr(self._arena.device)), "vu_cast_bf16")
        self._shadow_version = ver

    # ---- execution -------------------------------------------------------------------------
    def _run_forward(self, x: torch.Tensor, training: bool, seed: int, salt: Optional[torch.Tensor] = None):
        B = x.shape[0]
        xc = x.detach().float().contiguous()
        y = torch.empty_like(xc)
        ws = self._workspace(B, training)
        self.refresh_shadow()
        self._gen += 1
        check(lib().vu_model_forward(C.byref(self._cfg), ptr(self._arena), ptr(self._shadow), ptr(self._bn), ptr(xc),
                                     ptr(y), ptr(ws), ws.numel(), B, 1 if training else 0, seed, ptr(salt),
                                     stream_ptr(x.device)), "vu_model_forward")
        if training:
            self._nbt_pending += 1      # added to the num_batches_tracked buffers when state_dict() is taken
        return y

    def _run_backward(self, dy: torch.Tensor, training: bool, seed: int, need_dx: bool, stage: int = 0,
                      salt: Optional[torch.Tensor] = None):
        B = dy.shape[0]
        dyc = dy.detach().float().contiguous()
        dx = torch.empty_like(dyc) if need_dx else None
        ws = self._ws
        check(lib().vu_model_backward(C.byref(self._cfg), ptr(self._arena), ptr(self._shadow), ptr(self._bn),
                                      ptr(self._garena), ptr(dyc), ptr(dx), ptr(ws), ws.numel(), B,
                                      1 if training else 0, seed, ptr(salt), stage, stream_ptr(dy.device)),
              "vu_model_backward")
        return dx

    def forward(self, X: torch.Tensor):
        # model.py:376: Resize(im_size) is the identity for inputs that are already im_size^2 (D4)
        if X.dim() != 4:
            raise ValueError("expected (B,C,H,W)")
        if X.shape[-1] != self.im_size or X.shape[-2] != self.im_size:
            X = torch.nn.functional.interpolate(X.float(), size=(self.im_size, self.im_size), mode="bilinear",
                                                antialias=True, align_corners=False)   # parity unpinned (D4)
        assert X.shape[1] == self.num_channels, "Num. channels must agree"
        self._ensure_flat()
        training = self.training
        seed = _next_seed() if training and (self.attn_drop > 0 or self.proj_drop > 0) else 0
        if self._step_seed is not None:
            seed = self._step_seed
        if torch.is_grad_enabled():
            self._link_grads()
            return _ModelFn.apply(X, self._anchor, self, training, seed)
        return self._run_forward(X, training, seed)


def ViT_UNet(depth, depth_te, size_bottleneck, preprocessing, num_patches, patch_size, num_channels=3,
             hidden_dim=128, num_heads=8, attn_drop=0., proj_drop=0., linear_drop=0., dtype=torch.float32,
             projection_dim=None, verbose=False, attn_operands="storage"):
    """The constructor surface of README.md:18-31 / ViT_UNet.ipynb:974-990 (spec decision D2):
    `num_patches` replaces `im_size` (im_size = sqrt(num_patches) * patch_size)."""
    e = int(round(math.sqrt(num_patches)))
    assert e * e == num_patches, "num_patches must be a perfect square"
    if projection_dim is not None:
        assert projection_dim == num_channels * patch_size ** 2, "projection_dim must equal C * patch_size^2"
    return HViT_UNet(depth, depth_te, size_bottleneck, preprocessing, e * patch_size, patch_size, num_channels,
                     hidden_dim, num_heads, attn_drop, proj_drop, linear_drop, verbose=verbose, dtype=dtype,
                     attn_operands=attn_operands)


_PRESETS = {   # model.py:438-485
    "lite": dict(depth=2, depth_te=1, size_bottleneck=2, preprocessing="conv", im_size=224, patch_size=16,
                 num_channels=3, hidden_dim=64, num_heads=4, attn_drop=0.2, proj_drop=0.2, linear_drop=0),
    "base": dict(depth=2, depth_te=2, size_bottleneck=2, preprocessing="conv", im_size=224, patch_size=32,
                 num_channels=3, hidden_dim=128, num_heads=8, attn_drop=0.2, proj_drop=0.2, linear_drop=0),
    "large": dict(depth=2, depth_te=4, size_bottleneck=4, preprocessing="conv", im_size=224, patch_size=32,
                  num_channels=3, hidden_dim=128, num_heads=8, attn_drop=0.2, proj_drop=0.2, linear_drop=0),
}


def get_vit_unet(model_string: str, verbose=False, dtype=torch.float32, **overrides):
    key = model_string.lower()
    if key not in _PRESETS:
        raise ValueError(f"Model string {model_string} not valid")
    kw = dict(_PRESETS[key])
    kw.update(overrides)
    return HViT_UNet(verbose=verbose, dtype=dtype, **kw)
