"""CPU oracle for the ViT-UNet forward/backward path.  TEST INFRASTRUCTURE ONLY.

This file is a builder-owned, functional (module-free) torch-CPU fp32 restatement of the
arithmetic of the reference's packaged model (`/root/reference/vit_unet/torch/model.py`).
It exists to CHECK the HIP path.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product path never does (the product fails
loudly when the HIP library is missing).

Parity status: PINNED.  `tests/golden/make_golden.py` drives the *unmodified* reference forward
(`HViT_UNet.forward`, model.py:372-435, assembled by a harness because the reference ctor is
broken at model.py:309) in the build container and commits input/output/gradient vectors under
`tests/golden/`; `tests/test_oracle_golden.py` checks this restatement against them.

Each function cites the reference lines it restates.  All tensors are torch CPU tensors; all math
is fp32 unless the caller passes fp64 tensors (the functions are dtype-agnostic, which the
tests use for an fp64 "truth" run).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "Config", "retile", "patchify", "unpatchify", "downsample", "upsample",
    "conv3x3_per_patch", "reattention", "feed_forward", "te_block", "skip_block", "forward",
    "param_shapes", "param_count", "make_weights", "keep_mask", "keep_mask_quad", "flash_shape", "mse_loss", "psnr",
    "ssim", "dice_loss", "resize_u8", "warp_affine_u8", "invert_affine", "shift_scale_rotate_matrix",
    "denoise_prepare", "adamw_step", "PRESETS",
]


# --------------------------------------------------------------------------------------------
# configuration (HViT_UNet ctor arguments, model.py:264-299)
# --------------------------------------------------------------------------------------------
@dataclass
class Config:
    depth: int
    depth_te: int
    size_bottleneck: int
    preprocessing: str
    im_size: int
    patch_size: int
    num_channels: int
    hidden_dim: int
    num_heads: int
    attn_drop: float = 0.0
    proj_drop: float = 0.0
    linear_drop: float = 0.0
    attn_operands: str = "storage"    # "e4m3": q, k, v rounded to OCP e4m3 before the attention products (BASELINE config 5)

    def __post_init__(self):
        # model.py:281-283
        assert self.patch_size % (2 ** self.depth) == 0
        assert self.patch_size // (2 ** self.depth) >= 4
        assert self.im_size % self.patch_size == 0

    @property
    def P(self) -> int:  # elements of the latent image
        return self.num_channels * self.im_size * self.im_size

    def level(self, l: int) -> Tuple[int, int, int, int]:
        """(N, D, hidden, patch) at level l  (model.py:302-307)."""
        n0 = (self.im_size // self.patch_size) ** 2
        d0 = self.num_channels * self.patch_size ** 2
        return n0 * 4 ** l, d0 // 4 ** l, self.hidden_dim // 2 ** l, self.patch_size // 2 ** l


PRESETS = {  # model.py:438-485
    "lite": dict(depth=2, depth_te=1, size_bottleneck=2, preprocessing="conv", im_size=224,
                 patch_size=16, num_channels=3, hidden_dim=64, num_heads=4,
                 attn_drop=0.2, proj_drop=0.2, linear_drop=0.0),
    "base": dict(depth=2, depth_te=2, size_bottleneck=2, preprocessing="conv", im_size=224,
                 patch_size=32, num_channels=3, hidden_dim=128, num_heads=8,
                 attn_drop=0.2, proj_drop=0.2, linear_drop=0.0),
    "large": dict(depth=2, depth_te=4, size_bottleneck=4, preprocessing="conv", im_size=224,
                  patch_size=32, num_channels=3, hidden_dim=128, num_heads=8,
                  attn_drop=0.2, proj_drop=0.2, linear_drop=0.0),
}


# --------------------------------------------------------------------------------------------
# a1-a4: re-tiling (model.py:8-53).  SURVEY App. A: token n = (y//s)*e + x//s,
# feature f = ch*s*s + (y%s)*s + x%s.  The image layout is the s = im special case.
# --------------------------------------------------------------------------------------------
def patchify(img: torch.Tensor, s: int) -> torch.Tensor:
    """(B,C,H,W) -> (B,(H/s)(W/s),C*s*s)   [patch() model.py:8-18 + flatten(-3,-1)]"""
    B, C, H, W = img.shape
    assert H % s == 0 and W % s == 0
    t = img.reshape(B, C, H // s, s, W // s, s).permute(0, 2, 4, 1, 3, 5)
    return t.reshape(B, (H // s) * (W // s), C * s * s)


def unpatchify(tok: torch.Tensor, C: int) -> torch.Tensor:
    """(B,N,C*s*s) -> (B,C,e*s,e*s)   [unflatten+unpatch model.py:20-35]"""
    B, N, D = tok.shape
    s = int(round(math.sqrt(D // C)))
    e = int(round(math.sqrt(N)))
    assert e * e == N and C * s * s == D
    t = tok.reshape(B, e, e, C, s, s).permute(0, 3, 1, 4, 2, 5)
    return t.reshape(B, C, e * s, e * s)


def retile(tok: torch.Tensor, C: int, s_out: int) -> torch.Tensor:
    return patchify(unpatchify(tok, C), s_out)


def downsample(tok: torch.Tensor, C: int) -> torch.Tensor:
    """model.py:39-45: same image, patch size halved."""
    s = int(round(math.sqrt(tok.shape[-1] // C)))
    return retile(tok, C, s // 2)


def upsample(tok: torch.Tensor, C: int) -> torch.Tensor:
    """model.py:47-53: same image, patch size doubled."""
    s = int(round(math.sqrt(tok.shape[-1] // C)))
    return retile(tok, C, s * 2)


# --------------------------------------------------------------------------------------------
# dropout masks: the HIP path draws its Bernoulli masks from a counter-based integer hash
# (csrc/vu_common.h: vu_keep).  The oracle replays exactly the same hash so train-mode outputs
# with dropout>0 can be compared element for element.  (Bit-parity with torch's CPU RNG is
# impossible, SURVEY §7 hard part 5.)
# --------------------------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def _splitmix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def stream_key(seed: int, stream: int) -> Tuple[int, int]:
    k = _splitmix64((seed & 0xFFFFFFFFFFFFFFFF) ^ _splitmix64(stream & 0xFFFFFFFFFFFFFFFF))
    return k & 0xFFFFFFFF, (k >> 32) & 0xFFFFFFFF


def keep_mask(numel: int, p: float, seed: int, stream: int) -> torch.Tensor:
    """Boolean keep mask for `numel` linearly indexed elements (True = kept)."""
    if p <= 0.0:
        return torch.ones(numel, dtype=torch.bool)
    thr = int(round(p * 65536.0))
    k0, k1 = stream_key(seed, stream)
    idx = np.arange(numel, dtype=np.uint64)
    w = idx >> np.uint64(1)
    lo = w & _M32
    hi = w >> np.uint64(32)
    x = (_mix32(lo ^ np.uint64(k0)) + hi * np.uint64(0x9E3779B9) + np.uint64(k1)) & _M32
    r = np.where((idx & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
    return torch.from_numpy(r >= np.uint64(thr))


def _quad_word(x: np.ndarray, k0: int, k1: int) -> np.ndarray:
    """csrc/vu_flash.hip vu_quad_word: two rounds of (24-bit multiply-add, xor-shift 16), uint32 arithmetic."""
    m24 = np.uint64(0xFFFFFF)
    a = (x.astype(np.uint64) ^ np.uint64(k0)) & _M32
    y = ((a & m24) * np.uint64(0xB5297B) + (a >> np.uint64(12))) & _M32
    y ^= y >> np.uint64(16)
    y = ((y & m24) * np.uint64(0x9E3779) + (y >> np.uint64(12))) & _M32
    y ^= y >> np.uint64(16)
    return (y + np.uint64(k1)) & _M32


def quad_threshold(p: float) -> int:
    """8-bit drop threshold of the quad scheme: the drop probability is quad_threshold(p) / 256."""
    return min(255, max(1, (int(p * 65536.0 + 0.5) + 128) >> 8))


_QUAD_HEAD_MUL = (0x1E3779, 0x35297B, 0x68E31D, 0x7FEB35, 0x42B2AF, 0x65EBCB, 0x27D4EB, 0x165667)      # odd, below 2^23 (one v_mad_u32_u24 each)


def _quad_head(base: np.ndarray, h: int) -> np.ndarray:
    """csrc/vu_flash.hip vu_quad_head: the third round, with the head's own multiplier."""
    y = ((base & np.uint64(0xFFFFFF)) * np.uint64(_QUAD_HEAD_MUL[h & 7]) + (base >> np.uint64(12))) & _M32
    return y ^ (y >> np.uint64(16))


def keep_mask_quad(B: int, H: int, rows: int, n: int, p: float, seed: int, stream: int) -> torch.Tensor:
    """Keep mask (B, H, rows, n) of the non-materialising attention form (csrc/vu_flash.hip): one two-round hash word per
    (sample, map row, 4 consecutive keys), shared by the heads; head h runs it through a third round with its own
    multiplier; 8 bits per key, kept when byte >= round(256 p); n % 4 == 0."""
    assert n % 4 == 0
    k0, k1 = stream_key(seed, stream)
    x = np.arange(B * rows * (n // 4), dtype=np.uint64)
    base = _quad_word(x, k0, k1)
    thr = np.uint64(quad_threshold(p))
    out = np.empty((B, H, rows, n), dtype=bool)
    for h in range(H):
        w = _quad_head(base, h)
        by = np.stack([((w >> np.uint64(8 * r)) & np.uint64(255)) >= thr for r in range(4)], axis=1)
        out[:, h] = by.reshape(B, rows, n)
    return torch.from_numpy(out)


FLASH_FILL_RULE = True      # False: mirror VU_ATTN_FLASH=1 (the recompute form wherever the shape is covered)


def flash_shape(B: int, N: int, D: int, h: int) -> bool:
    """Mirror of vu_flash_ok and vu_flash_pays (csrc/vu_flash.hip) for bf16 storage: the shapes the model path runs in
    the non-materialising form by default, whose attention-map dropout uses the quad scheme."""
    if h <= 0 or D % h:
        return False
    d = D // h
    inst = (h == 8 and d in (24, 8, 32)) or (h == 4 and d in (32, 16, 12, 48))      # (d = 12: Lite's finest level, run as d = 16 on zero-padded operands; d = 48: Lite level 1)
    fills = (not FLASH_FILL_RULE) or B * ((N // 16 + 3) // 4) >= 192
    return inst and fills and N % 16 == 0 and N >= 256 and B * h * N * N < 2 ** 34


def _dropout_quad(x: torch.Tensor, p: float, training: bool, seed: Optional[int], stream: int):
    if (not training) or p <= 0.0:
        return x
    if seed is None:
        return F.dropout(x, p, True)
    B, H, rows, n = x.shape
    m = keep_mask_quad(B, H, rows, n, p, seed, stream).to(x.dtype)
    return x * m * (256.0 / (256.0 - quad_threshold(p)))


def _dropout(x: torch.Tensor, p: float, training: bool, seed: Optional[int], stream: int, row_pad: int = 1):
    """`row_pad` > 1: the last dimension is indexed with its length rounded up to a multiple of
    row_pad (the HIP attention maps are stored with 8-element-aligned rows and the mask index of
    element (row, j) is row * ld + j)."""
    if (not training) or p <= 0.0:
        return x
    if seed is None:   # timing-only path (bench.py cpu_baseline): torch's own Bernoulli, as the reference
        return F.dropout(x, p, True)
    n = x.shape[-1]
    ld = (n + row_pad - 1) // row_pad * row_pad
    rows = x.numel() // n
    m = keep_mask(rows * ld, p, seed, stream).reshape(rows, ld)[:, :n].reshape(x.shape).to(x.dtype)
    return x * m / (1.0 - p)


# --------------------------------------------------------------------------------------------
# bf16-storage emulation: the HIP path in bf16 mode computes in fp32 and rounds to bf16 exactly
# where a tensor is written to HBM.  `_r(t, st)` rounds at those points (straight-through in
# autograd) so a bf16 run can be checked against an oracle that follows the same rounding.
# --------------------------------------------------------------------------------------------
def _r(t: torch.Tensor, st) -> torch.Tensor:
    if st is None or st == torch.float32:
        return t
    return t + (t.detach().to(st).to(t.dtype) - t.detach())


# --------------------------------------------------------------------------------------------
# fp8 attention operands (BASELINE config 5; no counterpart in the reference, which has no reduced-precision path).
# OCP 8-bit floating point, E4M3 ("e4m3fn"): 1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities; largest
# finite value 448 = 1.75 * 2^8; smallest normal 2^-6; subnormals are multiples of 2^-9.  Conversion: round to nearest
# even, saturating.  Pinned by tests/test_oracle_golden.py against torch.float8_e4m3fn and the format's table values.
# --------------------------------------------------------------------------------------------
def round_e4m3(t: torch.Tensor) -> torch.Tensor:
    x = t.detach().to(torch.float64)
    a = x.abs().clamp(max=448.0)
    _, e = torch.frexp(a)                                   # a = m 2^e, m in [0.5, 1): binade exponent e - 1
    q = torch.pow(2.0, (e - 1).clamp(min=-6).to(torch.float64) - 3)      # spacing of the e4m3 grid around a
    r = torch.round(a / q) * q                              # torch.round: halves to even
    r = torch.where(torch.isnan(x), x, torch.copysign(r.clamp(max=448.0), x))
    return r.to(t.dtype)


def _e4(t: torch.Tensor, operands: str) -> torch.Tensor:
    """operand rounding, straight-through in autograd (the HIP path rounds q, k, v in place where the convolution
    wrote them and back-propagates as if it had not)"""
    if operands == "storage":
        return t
    assert operands == "e4m3", operands
    return t + (round_e4m3(t) - t.detach())


# --------------------------------------------------------------------------------------------
# K4: per-patch 3x3 convolution (model.py:137-139,152-154): zero halo at the PATCH border.
# --------------------------------------------------------------------------------------------
def conv3x3_per_patch(tok: torch.Tensor, C: int, w: torch.Tensor, b: Optional[torch.Tensor] = None):
    B, N, D = tok.shape
    s = int(round(math.sqrt(D // C)))
    y = F.conv2d(tok.reshape(B * N, C, s, s), w, b, padding=w.shape[-1] // 2)   # (1 x 1 kernels: the notebook variant)
    return y.reshape(B, N, D)


# --------------------------------------------------------------------------------------------
# a7/a9: Re-attention (model.py:150-164, 244-259).  `xq` feeds q, `xkv` feeds k and v
# (xq is xkv inside a transformer block; the skip connection passes encoder / decoder tensors).
# --------------------------------------------------------------------------------------------
def reattention(xq, xkv, p: Dict[str, torch.Tensor], pre: str, h: int, C: int, *, training: bool,
                attn_drop: float, proj_drop: float, seed: Optional[int] = None, stream: int = 0,
                bn_momentum: float = 0.1, eps: float = 1e-5, return_map: bool = False, storage=None,
                round_out: bool = True, flash: Optional[bool] = None, operands: str = "storage"):
    """`flash`: follow the rounding points / dropout scheme of the non-materialising HIP form (csrc/vu_flash.hip: the
    probabilities and the mixed map are never rounded to the storage type, only A^ as the PV operand; quad dropout
    scheme).  Default: what the model path runs - that form for bf16 storage on the shapes it covers."""
    B, N, D = xq.shape
    d = D // h
    st = storage
    if flash is None:
        flash = (storage == torch.bfloat16) and flash_shape(B, N, D, h)
    if flash:
        sm = None          # logits / probabilities stay fp32 in registers
    else:
        sm = st
    q = _e4(_r(conv3x3_per_patch(xq, C, p[pre + "qconv2d.weight"]), st), operands).reshape(B, N, h, d).permute(0, 2, 1, 3)
    k = _e4(_r(conv3x3_per_patch(xkv, C, p[pre + "kconv2d.weight"]), st), operands).reshape(B, N, h, d).permute(0, 2, 1, 3)
    v = _e4(_r(conv3x3_per_patch(xkv, C, p[pre + "vconv2d.weight"]), st), operands).reshape(B, N, h, d).permute(0, 2, 1, 3)
    s = _r(torch.matmul(q, k.transpose(-2, -1)) * (d ** -0.5), sm)    # model.py:155
    a = _r(torch.softmax(s, dim=-1), sm)                              # :156
    if flash:
        a = _dropout_quad(a, attn_drop, training, seed, 2 * stream)  # :157 (quad scheme)
    else:
        a = _dropout(a, attn_drop, training, seed, 2 * stream, row_pad=8)   # :157
    w = p[pre + "reatten_matrix.weight"].reshape(h, h)               # 1x1 conv across heads :159
    a = torch.einsum("gh,bhij->bgij", w, a) + p[pre + "reatten_matrix.bias"].reshape(1, h, 1, 1)
    gam, bet = p[pre + "var_norm.weight"], p[pre + "var_norm.bias"]
    if training:                                                      # BatchNorm2d train: batch stats
        mean = a.mean(dim=(0, 2, 3))
        var = a.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():                                         # running stats (momentum .1, unbiased var)
            n = a.numel() // h
            rm, rv = p[pre + "var_norm.running_mean"], p[pre + "var_norm.running_var"]
            rm.mul_(1 - bn_momentum).add_(bn_momentum * mean.detach().to(rm.dtype))
            rv.mul_(1 - bn_momentum).add_(bn_momentum * (var.detach() * n / max(n - 1, 1)).to(rv.dtype))
    else:
        mean, var = p[pre + "var_norm.running_mean"].to(a.dtype), p[pre + "var_norm.running_var"].to(a.dtype)
    a = (a - mean.reshape(1, h, 1, 1)) * torch.rsqrt(var.reshape(1, h, 1, 1) + eps)
    a = _r(a * gam.reshape(1, h, 1, 1) + bet.reshape(1, h, 1, 1), st)  # reatten_scale == 1.0 (:140)
    o = _r(torch.matmul(a, v).transpose(1, 2).reshape(B, N, D), st)   # :161
    y = F.linear(o, _r(p[pre + "proj.weight"], st), p[pre + "proj.bias"])     # :162
    y = _dropout(y, proj_drop, training, seed, 2 * stream + 1)                # :163
    if round_out:   # inside a block the HIP projection epilogue adds the residual before storing
        y = _r(y, st)
    return (y, a) if return_map else y


def _layernorm_nd(x, w, b, eps=1e-5):
    """LayerNorm(normalized_shape=(N,D)) (model.py:193-196): statistics over all N*D elements."""
    return F.layer_norm(x, x.shape[1:], w, b, eps)


def te_block(x, p, pre: str, cfg: Config, *, training: bool, seed=None, stream: int = 0, storage=None):
    """ReAttentionTransformerEncoder.forward (model.py:201-207), post-norm."""
    st = storage
    a = reattention(x, x, p, pre + "ReAttn.", cfg.num_heads, cfg.num_channels, training=training,
                    attn_drop=cfg.attn_drop, proj_drop=cfg.proj_drop, seed=seed, stream=stream, storage=st,
                    round_out=False, operands=cfg.attn_operands)
    x = _r(_layernorm_nd(_r(a + x, st), p[pre + "LN1.weight"], p[pre + "LN1.bias"]), st)
    f = feed_forward(x, p, pre + "FeedForward.", training=training, linear_drop=cfg.linear_drop, seed=seed, stream=stream,
                     storage=st)
    return _r(_layernorm_nd(_r(f + x, st), p[pre + "LN2.weight"], p[pre + "LN2.bias"]), st)


FF_STREAM = 1 << 32   # dropout streams of the two FeedForward sites: FF_STREAM + 2*stream (+1)  (csrc/vu_model.hip)


def feed_forward(x, p, pre: str, *, training: bool, linear_drop: float = 0.0, seed=None, stream: int = 0, storage=None):
    """FeedForward.forward (model.py:95-110): Linear -> GELU (exact erf) -> Dropout -> Linear -> Dropout.
    (linear_drop is 0 in every preset, model.py:451,467,483.)  The result is NOT rounded: inside a block the HIP
    epilogue adds the residual before storing."""
    st = storage
    hdn = F.gelu(F.linear(x, _r(p[pre + "net.0.weight"], st), p[pre + "net.0.bias"]))
    hdn = _r(_dropout(hdn, linear_drop, training, seed, FF_STREAM + 2 * stream), st)
    f = F.linear(hdn, _r(p[pre + "net.3.weight"], st), p[pre + "net.3.bias"])
    return _dropout(f, linear_drop, training, seed, FF_STREAM + 2 * stream + 1)


def skip_block(enc, dec, p, pre: str, cfg: Config, *, training: bool, seed=None, stream: int = 0, storage=None):
    """SkipConnection.forward(q=enc, k=dec, v=dec) (model.py:244-259): replaces dec, no residual."""
    assert enc.shape == dec.shape
    return reattention(enc, dec, p, pre, cfg.num_heads, cfg.num_channels, training=training,
                       attn_drop=cfg.attn_drop, proj_drop=cfg.proj_drop, seed=seed, stream=stream, storage=storage,
                       operands=cfg.attn_operands)


# --------------------------------------------------------------------------------------------
# a10: HViT_UNet.forward (model.py:372-435).  `stream` numbering of dropout sites follows
# execution order: Encoders, BottleNeck, then Decoders interleaved with SkipConnections.
# --------------------------------------------------------------------------------------------
def forward(p: Dict[str, torch.Tensor], cfg: Config, X: torch.Tensor, *, training: bool = False,
            seed: Optional[int] = None, taps: Optional[dict] = None, storage=None) -> torch.Tensor:
    B, C, H, W = X.shape
    assert C == cfg.num_channels and H == cfg.im_size and W == cfg.im_size  # D4: Resize == identity
    # PatchEncoder.forward (model.py:84-91): tokens + positional embedding (the unpatch/patch
    # round trip at :88-90 is a no-op).
    x = _r(patchify(X, cfg.patch_size) + p["PE.position_embedding.weight"].unsqueeze(0), storage)
    stream = 0
    skips: List[torch.Tensor] = []
    def tap_block(pre, xin, lvl):
        if taps is not None:   # teacher-forcing hooks for the tests: every block's input, level and dropout stream
            taps.setdefault("blocks", []).append((pre, xin.detach(), lvl, stream))
    for i in range(cfg.depth * cfg.depth_te):                          # :388-392
        tap_block(f"Encoders.{i}.", x, i // cfg.depth_te)
        x = te_block(x, p, f"Encoders.{i}.", cfg, training=training, seed=seed, stream=stream, storage=storage)
        stream += 1
        if (i + 1) % cfg.depth_te == 0:
            skips.append(x)
            x = downsample(x, C)
    if taps is not None:
        taps["after_encoders"] = x
    for i in range(cfg.size_bottleneck):                               # :400-401
        tap_block(f"BottleNeck.{i}.", x, cfg.depth)
        x = te_block(x, p, f"BottleNeck.{i}.", cfg, training=training, seed=seed, stream=stream, storage=storage)
        stream += 1
    if taps is not None:
        taps["after_bottleneck"] = x
    for i in range(cfg.depth * cfg.depth_te):                          # :410-418
        tap_block(f"Decoders.{i}.", x, cfg.depth - i // cfg.depth_te)
        x = te_block(x, p, f"Decoders.{i}.", cfg, training=training, seed=seed, stream=stream, storage=storage)
        stream += 1
        if (i + 1) % cfg.depth_te == 0:
            j = (i + 1) // cfg.depth_te
            x = upsample(x, C)
            enc = skips[cfg.depth - j]
            assert enc.shape == x.shape                                # :417
            if taps is not None:
                taps.setdefault("skips", []).append((f"SkipConnections.{j - 1}.", enc.detach(), x.detach(),
                                                     cfg.depth - j, stream))
            x = skip_block(enc, x, p, f"SkipConnections.{j - 1}.", cfg, training=training,
                           seed=seed, stream=stream, storage=storage)
            stream += 1
    if taps is not None:
        taps["after_decoders"] = x
    Y = unpatchify(x, C)                                               # :425
    if cfg.preprocessing == "conv":                                    # :427-428
        Y = F.conv2d(Y, p["conv2d.weight"], p["conv2d.bias"], padding=1)
    elif cfg.preprocessing == "fourier":
        raise NotImplementedError("D5: the reference 'fourier' branch (model.py:429-430) is a bug")
    return Y


# --------------------------------------------------------------------------------------------
# f4: the variants sketched in ViT_UNet.ipynb (classes PatchEncoder, FformerEncoder, ReAttention,
# ReAttentionTransformerEncoder, SkipConnection, ViT_UNet of the notebook).  The notebook does not run as committed
# (its Unpatch reads a notebook global), so there is no reference output to pin these against: "parity unpinned".
# They follow the notebook's text, with the re-tiling semantics of model.py:8-53.  `p` uses the module names of
# vit_unet/torch/variants.py (PE.conv2d, PE.position_embedding, <block>.ReAttn.*, <block>.LN, <block>.FeedForward.net.*).
# --------------------------------------------------------------------------------------------
def fft2_real(x: torch.Tensor) -> torch.Tensor:
    """`torch.fft.fft2(x).real` over the last two axes (FformerEncoder.forward; PatchEncoder 'fourier')."""
    return torch.fft.fft2(x).real


def fformer_block(x, p, pre: str, *, training: bool, linear_drop: float = 0.0, seed=None, stream: int = 0, storage=None):
    """FformerEncoder.forward: x += Re(fft2 x); x = LN(x); x += FF(x); x = LN(x), the SAME LayerNorm twice."""
    st = storage
    w, b = p[pre + "LN.weight"], p[pre + "LN.bias"]
    x = _r(_layernorm_nd(_r(_r(fft2_real(x), st) + x, st), w, b), st)
    f = feed_forward(x, p, pre + "FeedForward.", training=training, linear_drop=linear_drop, seed=seed, stream=stream, storage=st)
    return _r(_layernorm_nd(_r(f + x, st), w, b), st)


def notebook_te_block(x, p, pre: str, cfg: Config, *, training: bool, seed=None, stream: int = 0, storage=None):
    """the notebook's ReAttentionTransformerEncoder.forward: one LayerNorm, 1 x 1 q/k/v kernels (taken from the weights)."""
    st = storage
    w, b = p[pre + "LN.weight"], p[pre + "LN.bias"]
    a = reattention(x, x, p, pre + "ReAttn.", cfg.num_heads, cfg.num_channels, training=training, attn_drop=cfg.attn_drop,
                    proj_drop=cfg.proj_drop, seed=seed, stream=stream, storage=st, round_out=False, flash=False)
    x = _r(_layernorm_nd(_r(a + x, st), w, b), st)
    f = feed_forward(x, p, pre + "FeedForward.", training=training, linear_drop=cfg.linear_drop, seed=seed, stream=stream,
                     storage=st)
    return _r(_layernorm_nd(_r(f + x, st), w, b), st)


def notebook_forward(p, cfg: Config, X, *, training: bool = False, seed=None, storage=None, block: str = "reattention"):
    """the notebook's ViT_UNet.forward: PatchEncoder (image conv / Re fft2, positional embedding at the finest patch
    size, re-tiled to patch_size), the U of blocks, SkipConnections indexed (i - 1) // depth_te, output conv."""
    C = cfg.num_channels
    st = storage
    if cfg.preprocessing == "conv":
        X = _r(F.conv2d(X, p["PE.conv2d.weight"], p["PE.conv2d.bias"], padding=1), st)
    elif cfg.preprocessing == "fourier":
        X = _r(fft2_real(X), st)
    s_f = cfg.patch_size // 2 ** cfg.depth
    x = _r(patchify(X, s_f) + p["PE.position_embedding.weight"].unsqueeze(0), st)
    x = retile(x, C, cfg.patch_size)

    def blk(pre, x, stream):
        if block == "fformer":
            return fformer_block(x, p, pre, training=training, linear_drop=cfg.linear_drop, seed=seed, stream=stream, storage=st)
        return notebook_te_block(x, p, pre, cfg, training=training, seed=seed, stream=stream, storage=st)
    stream = 0
    skips: List[torch.Tensor] = []
    for i in range(cfg.depth * cfg.depth_te):
        x = blk(f"Encoders.{i}.", x, stream)
        stream += 1
        if (i + 1) % cfg.depth_te == 0:
            skips.append(x)
            x = downsample(x, C)
    for i in range(cfg.size_bottleneck):
        x = blk(f"BottleNeck.{i}.", x, stream)
        stream += 1
    for i in range(cfg.depth * cfg.depth_te):
        x = blk(f"Decoders.{i}.", x, stream)
        stream += 1
        if (i + 1) % cfg.depth_te == 0:
            x = upsample(x, C)
            enc = skips[cfg.depth - (i + 1) // cfg.depth_te]
            assert enc.shape == x.shape
            j = ((i - 1) // cfg.depth_te) % cfg.depth          # python list indexing of the notebook, negative index included
            x = reattention(enc, x, p, f"SkipConnections.{j}.", cfg.num_heads, C, training=training, attn_drop=cfg.attn_drop,
                            proj_drop=cfg.proj_drop, seed=seed, stream=stream, storage=st, flash=False)
            stream += 1
    Y = unpatchify(x, C)
    if cfg.preprocessing == "conv":
        Y = F.conv2d(Y, p["conv2d.weight"], p["conv2d.bias"], padding=1)
    return Y


# --------------------------------------------------------------------------------------------
# f4, second half: the layers of the reference's Keras re-implementation (/root/reference/vit_unet/tf/functions.py,
# tf/model.py).  TensorFlow is not in this image and the reference holds no outputs of these layers: "parity unpinned" -
# these functions restate the reference TEXT with torch CPU ops.  Conventions of vit_unet/torch/variants_tf.py: tokens
# (B, N, P) in this repository's channel-major feature order, Dense weights torch-style (out, in), parameter names = that
# module tree's state_dict keys.  Dropout masks: keep_mask with the linear element index (softmax rows: row * ld + j,
# ld = n rounded up to 8), streams 2 s (attention probabilities) and FF_STREAM + 2 s (+ 1) (FeedForward).
# --------------------------------------------------------------------------------------------
def tf_dense(x, p, pre, st=None):
    return F.linear(x, _r(p[pre + "weight"], st), p[pre + "bias"])


def tf_token_layernorm(x, p, pre, eps: float = 1e-3):
    """keras LayerNormalization(): axis -1, epsilon 1e-3 (tf/functions.py:281-282)."""
    return F.layer_norm(x, x.shape[-1:], p[pre + "weight"], p[pre + "bias"], eps)


def tf_feed_forward(x, p, pre, *, dropout: float, training: bool, seed=None, stream: int = 0, storage=None):
    """tf/functions.py:163-182: Dense -> gelu -> Dropout -> Dense -> gelu -> Dropout."""
    st = storage
    h = _r(F.gelu(_r(tf_dense(x, p, pre + "D1.", st), st)), st)
    h = _r(_dropout(h, dropout, training, seed, FF_STREAM + 2 * stream), st)
    f = _r(F.gelu(_r(tf_dense(h, p, pre + "D2.", st), st)), st)
    return _r(_dropout(f, dropout, training, seed, FF_STREAM + 2 * stream + 1), st)


def keras_mha(query, value, p, pre, *, num_heads: int, dropout: float, training: bool, seed=None, stream: int = 0, storage=None):
    """keras MultiHeadAttention(num_heads, key_dim)(query, value) with key = value (tf/functions.py:288, :389)."""
    st = storage
    B, Nq, _ = query.shape
    Nk = value.shape[1]
    q = _r(tf_dense(query, p, pre + "query.", st), st)
    k = _r(tf_dense(value, p, pre + "key.", st), st)
    v = _r(tf_dense(value, p, pre + "value.", st), st)
    kd = q.shape[-1] // num_heads
    qh = q.reshape(B, Nq, num_heads, kd).permute(0, 2, 1, 3)
    kh = k.reshape(B, Nk, num_heads, kd).permute(0, 2, 1, 3)
    vh = v.reshape(B, Nk, num_heads, kd).permute(0, 2, 1, 3)
    s = _r(torch.matmul(qh, kh.transpose(-2, -1)), st)
    a = _r(torch.softmax(s * (kd ** -0.5), dim=-1), st)
    a = _r(_dropout(a, dropout, training, seed, 2 * stream, row_pad=8), st)
    o = _r(torch.matmul(a, vh).permute(0, 2, 1, 3).reshape(B, Nq, num_heads * kd), st)
    return _r(tf_dense(o, p, pre + "output.", st), st)


def tf_attention_te(x, p, pre, *, layers: int, num_heads: int, attn_drop: float, proj_drop: float, training: bool, seed=None,
                    stream: int = 0, storage=None):
    """AttentionTransformerEncoder.call (tf/functions.py:302-311)."""
    st = storage
    for i in range(layers):
        a = keras_mha(x, x, p, f"{pre}Attn.{i}.", num_heads=num_heads, dropout=attn_drop, training=training, seed=seed,
                      stream=stream + i, storage=st)
        x = _r(tf_token_layernorm(_r(a + x, st), p, f"{pre}LN1.{i}."), st)
        f = tf_feed_forward(x, p, f"{pre}FF.{i}.", dropout=proj_drop, training=training, seed=seed, stream=stream + i, storage=st)
        x = _r(tf_token_layernorm(_r(f + x, st), p, f"{pre}LN2.{i}."), st)
    return x


def tf_token_pool4(x, mode: str):
    """Resampling 'max' / 'avg' before the position embedding (tf/functions.py:101-124), followed step by step:
    pool pairs of consecutive tokens; reshape (B, N/2, P) -> (B, N/4, 2, P); transpose to (B, 2, N/4, P); inside a map over
    the batch pool pairs along N/4; transpose back; concatenate the two slices of the middle axis along the tokens."""
    B, N, P = x.shape
    pool = (lambda t: torch.maximum(t[:, 0::2], t[:, 1::2])) if mode == "max" else (lambda t: 0.5 * (t[:, 0::2] + t[:, 1::2]))
    t1 = pool(x)                                            # (B, N/2, P)
    t3 = t1.reshape(B, N // 4, 2, P).permute(0, 2, 1, 3)    # (B, 2, N/4, P)
    t4 = pool(t3.reshape(B * 2, N // 4, P)).reshape(B, 2, N // 8, P)
    t5 = t4.permute(0, 2, 1, 3)                             # (B, N/8, 2, P)
    return torch.cat([t5[:, :, 0], t5[:, :, 1]], dim=-2)    # (B, N/4, P)


def tf_resampling(x, p, pre, *, kind: str, img_size: int, patch_size, C: int, storage=None):
    """Resampling.call (tf/functions.py:100-132); patch_size = [from, to]."""
    st = storage
    pos = p[pre + "position_embedding.weight"]
    if kind in ("max", "avg"):
        return _r(tf_token_pool4(x, kind) + pos, st)
    if kind == "standard":
        y = retile(x, C, patch_size[1])
        return _r(_r(tf_dense(y, p, pre + "linear.", st), st) + _r(pos, st), st)
    assert kind == "conv"
    B, N0, P0 = x.shape
    s = int(round(math.sqrt(P0 // C)))
    t = x.reshape(B, N0, C, s, s).permute(0, 2, 1, 3, 4).reshape(B * C, N0, s, s)         # patches are the channels
    t = _r(F.conv2d(t, _r(p[pre + "conv.weight"], st), p[pre + "conv.bias"], stride=2), st)  # kernel 2, stride 2: 'same' pads nothing
    N1 = t.shape[1]
    t = t.reshape(B, C, N1, s // 2, s // 2).permute(0, 2, 1, 3, 4).reshape(B, N1, P0 // 4)
    return _r(_r(tf_dense(t, p, pre + "linear.", st), st) + _r(pos, st), st)


def tf_patch_encoder(X, p, pre, *, patch: int, storage=None):
    """PatchEncoder.call (tf/functions.py:155-160)."""
    st = storage
    return _r(_r(tf_dense(patchify(X, patch), p, pre + "projection.", st), st) + _r(p[pre + "position_embedding.weight"], st), st)


def tf_forward(p, X, *, img_size: int, patch_size, num_channels: int, num_heads: int, transformer_layers, size_bottleneck: int,
               drop_attn: float, drop_proj: float, resampling_type: str, training: bool = False, seed=None, storage=None):
    """HViT_UNet.call of the Keras model with original_attn=True (tf/model.py:188-209)."""
    st = storage
    L = len(patch_size)
    rev = list(patch_size)[::-1]
    enc = tf_patch_encoder(_r(X, st), p, "PE.", patch=patch_size[0], storage=st)
    stream, skips = 0, []
    kw = dict(num_heads=num_heads, attn_drop=drop_attn, proj_drop=drop_proj, training=training, seed=seed, storage=st)
    for i in range(L - 1):
        enc = tf_attention_te(enc, p, f"Encoder.{i}.", layers=transformer_layers[i], stream=stream, **kw)
        stream += transformer_layers[i]
        skips.append(enc)
        enc = tf_resampling(enc, p, f"Encoder_RS.{i}.", kind=resampling_type, img_size=img_size, patch_size=list(patch_size[i:i + 2]),
                            C=num_channels, storage=st)
    enc = tf_attention_te(enc, p, "BottleNeck.", layers=size_bottleneck, stream=stream, **kw)
    stream += size_bottleneck
    skips = skips[::-1]
    for i in range(L - 1):
        enc = tf_resampling(enc, p, f"Decoder_RS.{i}.", kind=resampling_type, img_size=img_size, patch_size=rev[i:i + 2], C=num_channels,
                            storage=st)
        lay = transformer_layers[L - (i + 2)]
        enc = tf_attention_te(enc, p, f"Decoder.{i}.", layers=lay, stream=stream, **kw)
        stream += lay
        enc = keras_mha(skips[i], enc, p, f"SkipConnections.{i}.Attn.", num_heads=num_heads, dropout=drop_attn, training=training,
                        seed=seed, stream=stream, storage=st)
        stream += 1
    return (_r(X, st) + _r(unpatchify(enc, num_channels), st)).float()                   # tf/model.py:208: input residual


# --------------------------------------------------------------------------------------------
# parameters: names / shapes in reference registration order (model.py:309-370) and the
# builder-owned deterministic weight generator used by the golden fixtures.
# --------------------------------------------------------------------------------------------
def _attn_shapes(pre: str, D: int, h: int, C: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [
        (pre + "reatten_matrix.weight", (h, h, 1, 1)), (pre + "reatten_matrix.bias", (h,)),
        (pre + "var_norm.weight", (h,)), (pre + "var_norm.bias", (h,)),
        (pre + "qconv2d.weight", (C, C, 3, 3)), (pre + "kconv2d.weight", (C, C, 3, 3)),
        (pre + "vconv2d.weight", (C, C, 3, 3)),
        (pre + "proj.weight", (D, D)), (pre + "proj.bias", (D,)),
    ]


def _block_shapes(pre: str, N: int, D: int, hid: int, h: int, C: int):
    out = _attn_shapes(pre + "ReAttn.", D, h, C)
    out += [(pre + "LN1.weight", (N, D)), (pre + "LN1.bias", (N, D)),
            (pre + "LN2.weight", (N, D)), (pre + "LN2.bias", (N, D)),
            (pre + "FeedForward.net.0.weight", (hid, D)), (pre + "FeedForward.net.0.bias", (hid,)),
            (pre + "FeedForward.net.3.weight", (D, hid)), (pre + "FeedForward.net.3.bias", (D,))]
    return out


def param_shapes(cfg: Config) -> List[Tuple[str, Tuple[int, ...]]]:
    h, C = cfg.num_heads, cfg.num_channels
    N0, D0, _, _ = cfg.level(0)
    out: List[Tuple[str, Tuple[int, ...]]] = [("PE.position_embedding.weight", (N0, D0))]
    idx = 0
    for lvl in range(cfg.depth):
        N, D, hid, _ = cfg.level(lvl)
        for _ in range(cfg.depth_te):
            out += _block_shapes(f"Encoders.{idx}.", N, D, hid, h, C)
            idx += 1
    N, D, hid, _ = cfg.level(cfg.depth)
    for i in range(cfg.size_bottleneck):
        out += _block_shapes(f"BottleNeck.{i}.", N, D, hid, h, C)
    dec: List[Tuple[str, Tuple[int, ...]]] = []
    skp: List[Tuple[str, Tuple[int, ...]]] = []
    idx = 0
    for lvl in range(cfg.depth):
        N, D, hid, _ = cfg.level(cfg.depth - lvl)
        for _ in range(cfg.depth_te):
            dec += _block_shapes(f"Decoders.{idx}.", N, D, hid, h, C)
            idx += 1
        _, Ds, _, _ = cfg.level(cfg.depth - lvl - 1)
        skp += _attn_shapes(f"SkipConnections.{lvl}.", Ds, h, C)
    out += dec + skp
    if cfg.preprocessing == "conv":
        out += [("conv2d.weight", (C, C, 3, 3)), ("conv2d.bias", (C,))]
    return out


def buffer_shapes(cfg: Config) -> List[Tuple[str, Tuple[int, ...]]]:
    h = cfg.num_heads
    out = []
    for name, _ in param_shapes(cfg):
        if name.endswith("var_norm.weight"):
            pre = name[: -len("weight")]
            out += [(pre + "running_mean", (h,)), (pre + "running_var", (h,)),
                    (pre + "num_batches_tracked", ())]
    return out


def param_count(cfg: Config) -> int:
    return sum(int(np.prod(s)) for _, s in param_shapes(cfg))


def make_weights(cfg: Config, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic, well-conditioned weights (NumPy PCG64 -> tensors).  Not the torch default
    initialisers: every tensor is drawn non-trivially (LN/BN affine != 1/0, biases != 0) so the
    fixtures exercise every term."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg):
        n = int(np.prod(shape))
        if name.endswith("position_embedding.weight"):
            a = rng.standard_normal(n) * 0.5
        elif name.endswith(("LN1.weight", "LN2.weight", "var_norm.weight")):
            a = 1.0 + 0.2 * rng.standard_normal(n)
        elif name.endswith(("LN1.bias", "LN2.bias", "var_norm.bias")):
            a = 0.1 * rng.standard_normal(n)
        elif name.endswith("reatten_matrix.weight"):
            a = (np.eye(shape[0]).reshape(-1) + 0.3 * rng.standard_normal(n))
        elif name.endswith("conv2d.weight"):                           # q/k/v/out 3x3 convs
            a = rng.standard_normal(n) * (1.0 / math.sqrt(9 * shape[1]))
        elif name.endswith(".weight"):                                 # Linear (out,in)
            a = rng.standard_normal(n) * (1.0 / math.sqrt(shape[-1]))
        else:                                                          # biases
            a = 0.05 * rng.standard_normal(n)
        p[name] = torch.from_numpy(a.reshape(shape)).to(dtype)
    for name, shape in buffer_shapes(cfg):
        if name.endswith("running_mean"):
            p[name] = torch.from_numpy(0.01 * rng.standard_normal(shape)).to(dtype)
        elif name.endswith("running_var"):
            p[name] = torch.from_numpy(1e-4 * (1.0 + rng.random(shape))).to(dtype)
        else:
            p[name] = torch.zeros((), dtype=torch.int64)
    return p


def make_batch(cfg: Config, B: int, seed: int = 1234, dtype=torch.float32):
    """SURVEY §8d synthetic batch: clean y ~ U[0,1), noisy x = clip(y + N(0, 0.1^2))."""
    g = torch.Generator().manual_seed(seed)
    y = torch.rand(B, cfg.num_channels, cfg.im_size, cfg.im_size, generator=g)
    x = (y + 0.1 * torch.randn(y.shape, generator=g)).clamp(0.0, 1.0)
    return x.to(dtype), y.to(dtype)


# --------------------------------------------------------------------------------------------
# a13 + f2: loss / optimizer / metrics
# --------------------------------------------------------------------------------------------
def mse_loss(out, target):
    """torch.nn.MSELoss() (run_denoising.py:80): mean over all elements."""
    return ((out - target) ** 2).mean()


def dice_loss(inp, target, smooth: float = 1.0):
    """README.md:91-101."""
    i, t = inp.reshape(-1), target.reshape(-1)
    return 1 - (2.0 * (i * t).sum() + smooth) / (i.sum() + t.sum() + smooth)


def psnr(target, out, data_range: float = 1.0):
    """functions.py:7-19 counterpart: per-image 10*log10(R^2/MSE) (skimage semantics)."""
    B = target.shape[0]
    mse = ((target.double() - out.double()) ** 2).reshape(B, -1).mean(dim=1)
    return 10.0 * torch.log10(data_range ** 2 / mse)


def ssim(target, out, data_range: float = 1.0, win: int = 7, K1: float = 0.01, K2: float = 0.03):
    """Per-image mean SSIM, channels averaged (README.md:85-89 names the metric; the reference has
    no implementation).  Third-party algorithm restated: scikit-image (unpinned in
    requirements.txt, absent here) `structural_similarity(im1, im2, data_range=R, channel_axis=0)`
    with its defaults - `scipy.ndimage.uniform_filter(size=win)` local moments, sample covariance
    (N/(N-1)), S = (2 ux uy + C1)(2 vxy + C2) / ((ux^2 + uy^2 + C1)(vx + vy + C2)), C1 = (K1 R)^2,
    C2 = (K2 R)^2, mean of S in float64 after cropping (win-1)//2 on every side.  PARITY UNPINNED
    (no reference fixture exists for it); float64 throughout."""
    from scipy.ndimage import uniform_filter
    X = np.asarray(target, dtype=np.float64)
    Y = np.asarray(out, dtype=np.float64)
    B, C = X.shape[:2]
    NP = win * win
    cov_norm = NP / (NP - 1.0)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    pad = (win - 1) // 2
    res = np.zeros(B)
    for b in range(B):
        acc = 0.0
        for c in range(C):
            x, y = X[b, c], Y[b, c]
            ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
            uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
            vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
            S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
            acc += S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean()
        res[b] = acc / C
    return torch.from_numpy(res)


# --------------------------------------------------------------------------------------------
# f3: host-side input pipeline (dataset.py:52-71 + run_denoising.py:52-59), third-party arithmetic
# restated: OpenCV (`cv2`, unpinned in requirements.txt, absent here) 8-bit resize / warpAffine
# fixed-point schemes and albumentations' Normalize / ShiftScaleRotate.  PARITY UNPINNED against
# cv2 / albumentations themselves (neither can be run here and the reference holds no fixture).
# --------------------------------------------------------------------------------------------
def _lin_coef(dsize: int, ssize: int):
    """cv2 resize INTER_LINEAR tables for 8-bit images: tap pair and 11-bit coefficient pair."""
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * (ssize / dsize) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo, hi = s < 0, s >= ssize - 1
    f[lo] = 0.0
    s[lo] = 0
    f[hi] = 0.0
    s[hi] = ssize - 1
    s1 = np.minimum(s + 1, ssize - 1)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, s1, a0, a1


def resize_u8(img: np.ndarray, im: int) -> np.ndarray:
    """cv2.resize(img, (im, im)) for an (H,W,C) uint8 image, default INTER_LINEAR (dataset.py:55-56):
    two-pass 11-bit fixed point; an exact 2x reduction is the 2x2 box mean (OpenCV switches
    INTER_LINEAR to its fast INTER_AREA there)."""
    H, W, C = img.shape
    if H == im and W == im:
        return img.copy()
    v = img.astype(np.int64)
    if H == 2 * im and W == 2 * im:
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _lin_coef(im, W)
    y0, y1, b0, b1 = _lin_coef(im, H)
    S = v[:, x0, :] * a0[None, :, None] + v[:, x1, :] * a1[None, :, None]        # (H, im, C)
    S0, S1 = S[y0], S[y1]
    out = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def shift_scale_rotate_matrix(im: int, angle: float, scale: float, dx: float, dy: float) -> np.ndarray:
    """albumentations ShiftScaleRotate's forward 2x3 matrix: cv2.getRotationMatrix2D(center, angle,
    scale) with center = (w/2 - 0.5, h/2 - 0.5), then the shift (dx*w, dy*h) added."""
    cx = cy = im / 2.0 - 0.5
    a = scale * math.cos(math.radians(angle))
    b = scale * math.sin(math.radians(angle))
    return np.array([[a, b, (1 - a) * cx - b * cy + dx * im], [-b, a, b * cx + (1 - a) * cy + dy * im]], dtype=np.float64)


def invert_affine(M: np.ndarray) -> np.ndarray:
    """The inversion cv2.warpAffine applies to a forward matrix."""
    M = np.asarray(M, dtype=np.float64).reshape(2, 3)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    i00, i01, i10, i11 = A11, M[0, 1] * (-D), M[1, 0] * (-D), A22
    b1 = -i00 * M[0, 2] - i01 * M[1, 2]
    b2 = -i10 * M[0, 2] - i11 * M[1, 2]
    return np.array([[i00, i01, b1], [i10, i11, b2]], dtype=np.float64)


def warp_affine_u8(img: np.ndarray, Minv: np.ndarray, nearest: bool) -> np.ndarray:
    """cv2.warpAffine(img, M, (w,h), INTER_LINEAR | INTER_NEAREST, BORDER_CONSTANT, 0) for a square
    (im,im,C) uint8 image, given the INVERSE matrix: 10-bit fixed-point coordinates; bilinear at 1/32
    sub-pixel with integer weights (sum 1024, round half up); taps outside the image are 0."""
    im, _, C = img.shape
    xs = np.arange(im, dtype=np.float64)
    ys = np.arange(im, dtype=np.float64)
    adx = np.rint(Minv[0, 0] * xs * 1024.0).astype(np.int64)[None, :]
    ady = np.rint(Minv[1, 0] * xs * 1024.0).astype(np.int64)[None, :]
    X0 = np.rint((Minv[0, 1] * ys + Minv[0, 2]) * 1024.0).astype(np.int64)[:, None]
    Y0 = np.rint((Minv[1, 1] * ys + Minv[1, 2]) * 1024.0).astype(np.int64)[:, None]
    pad = np.zeros((im + 2, im + 2, C), dtype=np.int64)
    pad[1:-1, 1:-1] = img

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < im) & (xx >= 0) & (xx < im)
        return pad[np.where(ok, yy, -1) + 1, np.where(ok, xx, -1) + 1]
    if nearest:
        return tap((Y0 + 512 + ady) >> 10, (X0 + 512 + adx) >> 10).astype(np.uint8)
    X, Y = (X0 + 16 + adx) >> 5, (Y0 + 16 + ady) >> 5
    sx, sy, ax, ay = X >> 5, Y >> 5, (X & 31)[..., None], (Y & 31)[..., None]
    acc = ((32 - ax) * (32 - ay) * tap(sy, sx) + ax * (32 - ay) * tap(sy, sx + 1)
           + (32 - ax) * ay * tap(sy + 1, sx) + ax * ay * tap(sy + 1, sx + 1) + 512) >> 10
    return acc.astype(np.uint8)


def denoise_prepare(noisy: np.ndarray, clean: np.ndarray, im: int, fwd: Optional[np.ndarray] = None,
                    mean: float = 0.456, std: float = 0.224):
    """One batch of DenoisingDataset items (dataset.py:52-71) under run_denoising.py's train
    (`fwd` = (B,2,3) ShiftScaleRotate matrices) or validation (`fwd=None`) transform.
    noisy / clean: (B,H,W,C) uint8 -> x, y: (B,C,im,im) float32 torch tensors.
    Normalize (albumentations, float32): (v - mean*255) * (1 / (std*255)) on the image only; the
    "mask" (clean) is warped with nearest interpolation and not normalised; then both `/255.`."""
    B = noisy.shape[0]
    m255 = np.float32(mean) * np.float32(255.0)
    rden = np.float32(1.0) / (np.float32(std) * np.float32(255.0))
    xs, ys = [], []
    for b in range(B):
        n, c = resize_u8(noisy[b], im), resize_u8(clean[b], im)
        if fwd is not None:
            Minv = invert_affine(fwd[b])
            n, c = warp_affine_u8(n, Minv, nearest=False), warp_affine_u8(c, Minv, nearest=True)
        xn = ((n.astype(np.float32) - m255) * rden) / np.float32(255.0)
        yc = (c / 255.0).astype(np.float32)
        xs.append(xn.transpose(2, 0, 1))
        ys.append(yc.transpose(2, 0, 1))
    return torch.from_numpy(np.stack(xs)), torch.from_numpy(np.stack(ys))


def _lin_coef_f(dsize: int, ssize: int):
    """cv2 resize INTER_LINEAR for non-8-bit types: float tap weights (no fixed point)."""
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= ssize - 1
    f[lo | hi] = 0.0
    s[lo] = 0
    s[hi] = ssize - 1
    return s, np.minimum(s + 1, ssize - 1), np.float32(1.0) - f, f, scale


def _sat_s16(v: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(v).astype(np.int64), -32768, 32767)


def seg_resize(img: Optional[np.ndarray], mask: Optional[np.ndarray], oh: int, ow: int):
    """cv2.resize of one SegmentationDataset item to (oh, ow): the int16 slice with INTER_LINEAR
    (float coefficients, float accumulation, round-half-even, saturate; exact 2x reduction = 2x2 box
    mean), the label mask with INTER_NEAREST (floor(x * scale) clamped)."""
    ref = img if img is not None else mask
    H, W = ref.shape
    if H == oh and W == ow:
        return (None if img is None else img.astype(np.int64)), (None if mask is None else mask.copy())
    x0, x1, a0, a1, sx = _lin_coef_f(ow, W)
    y0, y1, b0, b1, sy = _lin_coef_f(oh, H)
    ri = rm = None
    if img is not None:
        v = img.astype(np.int64)
        if H == 2 * oh and W == 2 * ow:
            ri = (v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2
        else:
            vf = img.astype(np.float32)
            S = vf[:, x0] * a0[None, :] + vf[:, x1] * a1[None, :]
            ri = _sat_s16(S[y0] * b0[:, None] + S[y1] * b1[:, None])
    if mask is not None:
        my = np.minimum(np.floor(np.arange(oh, dtype=np.float64) * sy).astype(np.int64), H - 1)
        mx = np.minimum(np.floor(np.arange(ow, dtype=np.float64) * sx).astype(np.int64), W - 1)
        rm = mask[my][:, mx]
    return ri, rm


def seg_warp(img: Optional[np.ndarray], mask: Optional[np.ndarray], Minv: np.ndarray):
    """cv2.warpAffine of an (oh, ow) int16 slice (INTER_LINEAR: 10-bit fixed-point coordinates, 1/32
    sub-pixel, FLOAT weights (1-fy)(1-fx).., sum left to right, round-half-even) and its mask
    (INTER_NEAREST), constant border 0, given the INVERSE matrix."""
    ref = img if img is not None else mask
    oh, ow = ref.shape
    xs, ys = np.arange(ow, dtype=np.float64), np.arange(oh, dtype=np.float64)
    adx = np.rint(Minv[0, 0] * xs * 1024.0).astype(np.int64)[None, :]
    ady = np.rint(Minv[1, 0] * xs * 1024.0).astype(np.int64)[None, :]
    X0 = np.rint((Minv[0, 1] * ys + Minv[0, 2]) * 1024.0).astype(np.int64)[:, None]
    Y0 = np.rint((Minv[1, 1] * ys + Minv[1, 2]) * 1024.0).astype(np.int64)[:, None]

    def tap(a, yy, xx):
        ok = (yy >= 0) & (yy < oh) & (xx >= 0) & (xx < ow)
        return np.where(ok, a[np.clip(yy, 0, oh - 1), np.clip(xx, 0, ow - 1)], 0)
    ri = rm = None
    if img is not None:
        X, Y = (X0 + 16 + adx) >> 5, (Y0 + 16 + ady) >> 5
        sx, sy = X >> 5, Y >> 5
        fx, fy = (X & 31).astype(np.float32) * np.float32(0.03125), (Y & 31).astype(np.float32) * np.float32(0.03125)
        gx, gy = np.float32(1.0) - fx, np.float32(1.0) - fy
        f = lambda a: a.astype(np.float32)
        acc = f(tap(img, sy, sx)) * (gy * gx)
        acc = acc + f(tap(img, sy, sx + 1)) * (gy * fx)
        acc = acc + f(tap(img, sy + 1, sx)) * (fy * gx)
        acc = acc + f(tap(img, sy + 1, sx + 1)) * (fy * fx)
        ri = _sat_s16(acc)
    if mask is not None:
        rm = tap(mask, (Y0 + 512 + ady) >> 10, (X0 + 512 + adx) >> 10)
    return ri, rm


def seg_prepare(image: Optional[np.ndarray], mask: Optional[np.ndarray], im_size, fwd: Optional[np.ndarray] = None,
                lo: float = -1024.0, hi: float = 1024.0, ls: float = 0.0):
    """One batch of SegmentationDataset items (dataset.py:18-38: DICOM slice `x`, NIfTI plane `mask`)
    through the scaling / augmentation the reference leaves to `augments`: resize to `im_size` ->
    optional ShiftScaleRotate (`fwd`: (B,2,3) forward matrices) -> x = clip((v-lo)/(hi-lo), 0, 1);
    y = label (1-ls) + ls/2.  image (B,H,W) int16, mask (B,H,W) uint8 -> (B,1,oh,ow) float32."""
    oh, ow = (im_size, im_size) if isinstance(im_size, int) else im_size
    B = (image if image is not None else mask).shape[0]
    xs, ys = [], []
    rng_ = np.float32(hi) - np.float32(lo)
    keep, floor_ = np.float32(1.0) - np.float32(ls), np.float32(0.5) * np.float32(ls)
    for b in range(B):
        ri, rm = seg_resize(None if image is None else image[b], None if mask is None else mask[b], oh, ow)
        if fwd is not None:
            ri, rm = seg_warp(ri, rm, invert_affine(fwd[b]))
        if ri is not None:
            xs.append(np.clip((ri.astype(np.float32) - np.float32(lo)) / rng_, np.float32(0), np.float32(1))[None])
        if rm is not None:
            ys.append((rm.astype(np.float32) * keep + floor_)[None])
    x = torch.from_numpy(np.stack(xs)) if xs else None
    y = torch.from_numpy(np.stack(ys)) if ys else None
    return x, y


def adamw_step(p, g, m, v, step: int, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2):
    """torch.optim.AdamW defaults (run_denoising.py:81), one tensor, in place.  `step` >= 1."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p
