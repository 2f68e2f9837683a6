"""SURVEY section 8 row f4, second half: the layers that exist only in the reference's Keras re-implementation
(`/root/reference/vit_unet/tf/functions.py`, `tf/model.py`), restated on this library's HIP entry points:

  * `Resampling` (tf/functions.py:60-132): 'max' / 'avg' token pooling, 'standard' (re-tiling + Dense), 'conv' (a strided
    2 x 2 convolution that mixes PATCHES as channels, then Dense) - each followed by the layer's position embedding;
  * `PatchEncoder` (:135-160): tokens -> Dense -> + position embedding;
  * `FeedForward` (:163-182): Dense -> GELU -> Dropout -> Dense -> GELU -> Dropout (a GELU after BOTH layers);
  * `AttentionTransformerEncoder` (:258-311): post-norm blocks of Keras `MultiHeadAttention` (key_dim = projection_dim per
    head), `LayerNormalization` over the last axis (epsilon 1e-3), the FeedForward above;
  * `SkipConnection` (:371-395): `MultiHeadAttention(query = encoder tensor, value = key = decoder tensor)`;
  * `HViT_UNet` (tf/model.py:9-209) with `original_attn=True`, including the input residual `Y = X + unpatch(...)` (:208).

TensorFlow is not in this image, so nothing here is pinned by a run of the reference: the CPU oracle under `oracle/` (its
`tf_*` functions) restates the same text with torch CPU ops and is the parity reference ("unpinned").  Conventions: tokens
are (B, N, P) with this repository's channel-major feature order (the Keras code is channels-last; the Dense / embedding
weights are simply indexed in this order), Dense weights are stored torch-style (out, in).  Not on the benchmarked path: the
ops are plain streaming kernels (csrc/vu_tfops.hip) and `vu_gemm`; nothing computes with torch apart from view / permute
copies and dtype casts at the model boundary.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .model import _RetileFn, _next_seed

nn = torch.nn

FF_STREAM = 1 << 32        # dropout streams: attention 2 s, FeedForward FF_STREAM + 2 s (+ 1), as in the main model


def _code(t):
    if t.dtype not in _lib.DTYPE_CODE:
        raise TypeError("float32 or bfloat16 tensors only")
    return _lib.DTYPE_CODE[t.dtype]


def _gemm(A, Bm, C, M, N, K, sAm, sAk, sBk, sBn, ldc, Z1=1, Z2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), bias=None, c_float=0):
    check(lib().vu_gemm(_code(A), c_float, ptr(A), ptr(Bm), ptr(C), M, N, K, sAm, sAk, sBk, sBn, ldc, Z1, Z2, sA[0], sA[1],
                        sB[0], sB[1], sC[0], sC[1], 1.0, ptr(bias) if bias is not None else None, 0, stream_ptr(A.device)), "vu_gemm")


def _as_storage(w: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 parameter -> the compute dtype (bf16 through vu_cast_bf16)."""
    wf = w.detach().float().contiguous()
    if dtype == torch.float32:
        return wf
    out = torch.empty(wf.shape, dtype=torch.bfloat16, device=wf.device)
    n = wf.numel()
    pad = (-n) % 4
    if pad:
        return wf.to(torch.bfloat16)
    check(lib().vu_cast_bf16(ptr(wf), ptr(out), n, stream_ptr(wf.device)), "vu_cast_bf16")
    return out


class _DenseFn(torch.autograd.Function):
    """y = x W^T + b over the last axis (keras Dense), W stored (out, in)."""

    @staticmethod
    def forward(ctx, x, W, b):
        xs = x.contiguous()
        K, N = xs.shape[-1], W.shape[0]
        M = xs.numel() // K
        Ws = _as_storage(W, xs.dtype)
        bf = b.detach().float().contiguous()
        y = torch.empty(*xs.shape[:-1], N, dtype=xs.dtype, device=xs.device)
        _gemm(xs, Ws, y, M, N, K, K, 1, 1, K, N, bias=bf)
        ctx.save_for_backward(xs, Ws)
        return y

    @staticmethod
    def backward(ctx, g):
        xs, Ws = ctx.saved_tensors
        gs = g.contiguous()
        K, N = xs.shape[-1], Ws.shape[0]
        M = xs.numel() // K
        dx = torch.empty_like(xs)
        _gemm(gs, Ws, dx, M, K, N, N, 1, K, 1, K)                                  # dx = g W
        dW = torch.zeros(N, K, dtype=torch.float32, device=xs.device)
        _gemm(gs, xs, dW, N, K, M, 1, N, K, 1, K, c_float=1)                        # dW = g^T x
        db = torch.zeros(N, dtype=torch.float32, device=xs.device)
        check(lib().vu_colsum(_code(gs), ptr(gs), ptr(db), M, N, N, stream_ptr(gs.device)), "vu_colsum")
        return dx, dW, db


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xs = x.contiguous()
        y = torch.empty_like(xs)
        check(lib().vu_gelu_fwd(_code(xs), ptr(xs), ptr(y), xs.numel(), stream_ptr(xs.device)), "vu_gelu_fwd")
        ctx.save_for_backward(xs)
        return y

    @staticmethod
    def backward(ctx, g):
        (xs,) = ctx.saved_tensors
        gs = g.contiguous()
        dx = torch.empty_like(xs)
        check(lib().vu_gelu_bwd(_code(xs), ptr(xs), ptr(gs), ptr(dx), xs.numel(), stream_ptr(xs.device)), "vu_gelu_bwd")
        return dx


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, stream_id):
        xs = x.contiguous()
        y = torch.empty_like(xs)
        check(lib().vu_dropout(_code(xs), ptr(xs), ptr(y), xs.numel(), p, seed, stream_id, stream_ptr(xs.device)), "vu_dropout")
        ctx.args = (p, seed, stream_id)
        return y

    @staticmethod
    def backward(ctx, g):
        p, seed, stream_id = ctx.args
        gs = g.contiguous()
        dx = torch.empty_like(gs)
        check(lib().vu_dropout(_code(gs), ptr(gs), ptr(dx), gs.numel(), p, seed, stream_id, stream_ptr(gs.device)), "vu_dropout")
        return dx, None, None, None


def _dropout(x, p, training, seed, stream_id):
    return _DropoutFn.apply(x, float(p), int(seed), int(stream_id)) if (training and p > 0.0) else x


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a_, b_ = a.contiguous(), b.contiguous()
        out = torch.empty_like(a_)
        check(lib().vu_add(_code(a_), ptr(a_), ptr(b_), ptr(out), a_.numel(), stream_ptr(a_.device)), "vu_add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class _TokenLayerNormFn(torch.autograd.Function):
    """LayerNormalization(axis=-1, epsilon) of a + x: per TOKEN statistics, affine (D,) - the library's residual + LayerNorm
    kernels with one "sample" per token (rows are processed in slabs of 32768: the kernels index the row with gridDim.y)."""
    SLAB = 32768

    @staticmethod
    def forward(ctx, a, x, w, b, eps):
        a_, x_ = a.contiguous(), x.contiguous()
        D = a_.shape[-1]
        rows = a_.numel() // D
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        z, y = torch.empty_like(a_), torch.empty_like(a_)
        stats = torch.empty(rows, 2, dtype=torch.float32, device=a_.device)
        L = lib()
        av, xv, zv, yv = a_.view(rows, D), x_.view(rows, D), z.view(rows, D), y.view(rows, D)
        for r0 in range(0, rows, _TokenLayerNormFn.SLAB):
            r1 = min(rows, r0 + _TokenLayerNormFn.SLAB)
            ws = torch.empty(L.vu_layernorm_workspace_floats(r1 - r0, D), dtype=torch.float32, device=a_.device)
            check(L.vu_add_layernorm_fwd_eps(_code(a_), ptr(av[r0:r1]), ptr(xv[r0:r1]), ptr(zv[r0:r1]), ptr(wf), ptr(bf), ptr(yv[r0:r1]),
                                             ptr(ws), ptr(stats[r0:r1]), r1 - r0, D, eps, stream_ptr(a_.device)), "vu_add_layernorm_fwd_eps")
        ctx.save_for_backward(z, wf, stats)
        return y

    @staticmethod
    def backward(ctx, g):
        z, wf, stats = ctx.saved_tensors
        gs = g.contiguous()
        D = z.shape[-1]
        rows = z.numel() // D
        dz = torch.empty_like(z)
        dw, db = torch.zeros_like(wf), torch.zeros_like(wf)
        L = lib()
        gv, zv, dv = gs.view(rows, D), z.view(rows, D), dz.view(rows, D)
        for r0 in range(0, rows, _TokenLayerNormFn.SLAB):
            r1 = min(rows, r0 + _TokenLayerNormFn.SLAB)
            ws = torch.empty(L.vu_layernorm_workspace_floats(r1 - r0, D), dtype=torch.float32, device=z.device)
            check(L.vu_layernorm_bwd(_code(z), ptr(gv[r0:r1]), ptr(zv[r0:r1]), ptr(wf), ptr(stats[r0:r1]), ptr(dw), ptr(db), ptr(ws),
                                     ptr(dv[r0:r1]), r1 - r0, D, stream_ptr(z.device)), "vu_layernorm_bwd")
        return dz, dz, dw, db, None


class _Pool4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, mode):
        xs = x.contiguous()
        B, N, P = xs.shape
        posf = pos.detach().float().contiguous()
        y = torch.empty(B, N // 4, P, dtype=xs.dtype, device=xs.device)
        check(lib().vu_token_pool4_fwd(_code(xs), mode, ptr(xs), ptr(posf), ptr(y), B, N, P, stream_ptr(xs.device)), "vu_token_pool4_fwd")
        ctx.save_for_backward(xs)
        ctx.mode, ctx.pshape = mode, pos.shape
        return y

    @staticmethod
    def backward(ctx, g):
        (xs,) = ctx.saved_tensors
        gs = g.contiguous()
        B, N, P = xs.shape
        dx = torch.zeros_like(xs)          # (every source token is written; zeros only guard against a future sparse mode)
        check(lib().vu_token_pool4_bwd(_code(xs), ctx.mode, ptr(xs), ptr(gs), ptr(dx), B, N, P, stream_ptr(xs.device)), "vu_token_pool4_bwd")
        dpos = torch.zeros((N // 4) * P, dtype=torch.float32, device=xs.device)
        check(lib().vu_colsum(_code(gs), ptr(gs), ptr(dpos), B, (N // 4) * P, (N // 4) * P, stream_ptr(gs.device)), "vu_colsum")
        return dx, dpos.reshape(ctx.pshape), None


class _AddPosFn(torch.autograd.Function):
    """x + position_embedding (broadcast over the batch); the embedding gradient is the column sum over the batch."""

    @staticmethod
    def forward(ctx, x, pos):
        xs = x.contiguous()
        pe = _as_storage(pos, xs.dtype).reshape(1, *xs.shape[1:]).expand_as(xs).contiguous()
        out = torch.empty_like(xs)
        check(lib().vu_add(_code(xs), ptr(xs), ptr(pe), ptr(out), xs.numel(), stream_ptr(xs.device)), "vu_add")
        ctx.pshape = pos.shape
        return out

    @staticmethod
    def backward(ctx, g):
        gs = g.contiguous()
        B = gs.shape[0]
        n = gs[0].numel()
        dpos = torch.zeros(n, dtype=torch.float32, device=gs.device)
        check(lib().vu_colsum(_code(gs), ptr(gs), ptr(dpos), B, n, n, stream_ptr(gs.device)), "vu_colsum")
        return g, dpos.reshape(ctx.pshape)


class _MhaCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(kd)) -> dropout -> . v, heads side by side in the feature axis: q (B,Nq,H kd), k / v (B,Nk,H kd)."""

    @staticmethod
    def forward(ctx, q, k, v, H, p, seed, stream_id):
        q_, k_, v_ = q.contiguous(), k.contiguous(), v.contiguous()
        B, Nq, HD = q_.shape
        Nk, kd = k_.shape[1], HD // H
        ld = (Nk + 7) // 8 * 8
        scale = 1.0 / math.sqrt(kd)
        dev, dt = q_.device, q_.dtype
        s = torch.zeros(B, H, Nq, ld, dtype=dt, device=dev)
        _gemm(q_, k_, s, Nq, Nk, kd, HD, 1, 1, HD, ld, B, H, (Nq * HD, kd), (Nk * HD, kd), (H * Nq * ld, Nq * ld))
        P, Pd = torch.zeros_like(s), torch.zeros_like(s)
        check(lib().vu_softmax_rows_fwd(_code(s), ptr(s), ptr(P), ptr(Pd), B * H * Nq, Nk, ld, scale, p, seed, stream_id, stream_ptr(dev)),
              "vu_softmax_rows_fwd")
        o = torch.empty(B, Nq, HD, dtype=dt, device=dev)
        _gemm(Pd, v_, o, Nq, kd, Nk, ld, 1, HD, 1, HD, B, H, (H * Nq * ld, Nq * ld), (Nk * HD, kd), (Nq * HD, kd))
        ctx.save_for_backward(q_, k_, v_, P, Pd)
        ctx.args = (H, p, seed, stream_id, ld, scale)
        return o

    @staticmethod
    def backward(ctx, g):
        q_, k_, v_, P, Pd = ctx.saved_tensors
        H, p, seed, stream_id, ld, scale = ctx.args
        go = g.contiguous()
        B, Nq, HD = q_.shape
        Nk, kd = k_.shape[1], HD // H
        dev, dt = q_.device, q_.dtype
        sA = (H * Nq * ld, Nq * ld)
        dPd = torch.zeros(B, H, Nq, ld, dtype=dt, device=dev)
        _gemm(go, v_, dPd, Nq, Nk, kd, HD, 1, 1, HD, ld, B, H, (Nq * HD, kd), (Nk * HD, kd), sA)               # dPd = do v^T
        dv = torch.empty_like(v_)
        _gemm(Pd, go, dv, Nk, kd, Nq, 1, ld, HD, 1, HD, B, H, sA, (Nq * HD, kd), (Nk * HD, kd))                # dv = Pd^T do
        ds = torch.zeros_like(dPd)
        check(lib().vu_softmax_rows_bwd(_code(P), ptr(P), ptr(dPd), ptr(ds), B * H * Nq, Nk, ld, scale, p, seed, stream_id, stream_ptr(dev)),
              "vu_softmax_rows_bwd")
        dq = torch.empty_like(q_)
        _gemm(ds, k_, dq, Nq, kd, Nk, ld, 1, HD, 1, HD, B, H, sA, (Nk * HD, kd), (Nq * HD, kd))                # dq = ds k
        dk = torch.empty_like(k_)
        _gemm(ds, q_, dk, Nk, kd, Nq, 1, ld, HD, 1, HD, B, H, sA, (Nq * HD, kd), (Nk * HD, kd))                # dk = ds^T q
        return dq, dk, dv, None, None, None, None


# ---------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------
class Dense(nn.Linear):
    """keras Dense over the last axis; parameters as torch.nn.Linear (weight (out, in), bias)."""

    def forward(self, x):
        return _DenseFn.apply(x, self.weight, self.bias)


class TokenLayerNorm(nn.Module):
    """keras LayerNormalization(): axis = -1, epsilon = 1e-3, gamma / beta of shape (D,)."""

    def __init__(self, dim: int, eps: float = 1e-3):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.zeros(dim)), eps

    def forward(self, a, x):
        return _TokenLayerNormFn.apply(a, x, self.weight, self.bias, self.eps)


class KerasMultiHeadAttention(nn.Module):
    """keras MultiHeadAttention(num_heads, key_dim, value_dim = key_dim, dropout), call(query, value) with key = value
    (tf/functions.py:288, :389): per-head projections of size key_dim, scores / sqrt(key_dim), softmax over the keys, dropout
    on the probabilities, value product, output projection back to the query's feature size."""

    def __init__(self, dim: int, num_heads: int, key_dim: int, dropout: float = 0.0):
        super().__init__()
        self.dim, self.num_heads, self.key_dim, self.dropout = dim, num_heads, key_dim, dropout
        self.query, self.key, self.value = Dense(dim, num_heads * key_dim), Dense(dim, num_heads * key_dim), Dense(dim, num_heads * key_dim)
        self.output = Dense(num_heads * key_dim, dim)

    def forward(self, query, value, seed=0, stream_id=0):
        q, k, v = self.query(query), self.key(value), self.value(value)
        p = self.dropout if self.training else 0.0
        o = _MhaCoreFn.apply(q, k, v, self.num_heads, float(p), int(seed), int(2 * stream_id))
        return self.output(o)


class FeedForward(nn.Module):
    """tf/functions.py:163-182: Dense -> GELU -> Dropout -> Dense -> GELU -> Dropout."""

    def __init__(self, projection_dim: int, hidden_dim: int, dropout: float):
        super().__init__()
        self.D1, self.D2, self.dropout = Dense(projection_dim, hidden_dim), Dense(hidden_dim, projection_dim), dropout

    def forward(self, x, seed=0, stream_id=0):
        h = _dropout(_GeluFn.apply(self.D1(x)), self.dropout, self.training, seed, FF_STREAM + 2 * stream_id)
        return _dropout(_GeluFn.apply(self.D2(h)), self.dropout, self.training, seed, FF_STREAM + 2 * stream_id + 1)


class AttentionTransformerEncoder(nn.Module):
    """tf/functions.py:258-311: `transformer_layers` post-norm blocks of Keras MHA + FeedForward."""

    def __init__(self, num_heads: int, transformer_layers: int, projection_dim: int, hidden_dim: int, attn_drop: float, proj_drop: float):
        super().__init__()
        self.LN1 = nn.ModuleList([TokenLayerNorm(projection_dim) for _ in range(transformer_layers)])
        self.LN2 = nn.ModuleList([TokenLayerNorm(projection_dim) for _ in range(transformer_layers)])
        self.Attn = nn.ModuleList([KerasMultiHeadAttention(projection_dim, num_heads, projection_dim, attn_drop) for _ in range(transformer_layers)])
        self.FF = nn.ModuleList([FeedForward(projection_dim, hidden_dim, proj_drop) for _ in range(transformer_layers)])

    def forward(self, x, seed=0, stream_id=0):
        for i in range(len(self.Attn)):
            a = self.Attn[i](x, x, seed=seed, stream_id=stream_id + i)
            x = self.LN1[i](a, x)
            f = self.FF[i](x, seed=seed, stream_id=stream_id + i)
            x = self.LN2[i](f, x)
        return x


class PatchEncoder(nn.Module):
    """tf/functions.py:135-160: tokens at `patch_size` -> Dense(projection_dim) -> + position embedding."""

    def __init__(self, img_size: int, patch_size: int, num_channels: int, projection_dim: Optional[int] = None):
        super().__init__()
        self.img_size, self.patch_size, self.num_channels = img_size, patch_size, num_channels
        self.num_patches = (img_size // patch_size) ** 2
        self.projection_dim = projection_dim if projection_dim is not None else num_channels * patch_size ** 2
        self.projection = Dense(num_channels * patch_size ** 2, self.projection_dim)
        self.position_embedding = nn.Embedding(self.num_patches, self.projection_dim)

    def forward(self, X):
        B, C_, im, _ = X.shape
        tok = _RetileFn.apply(X.reshape(B, 1, C_ * im * im), C_, im, im, self.patch_size)
        return _AddPosFn.apply(self.projection(tok), self.position_embedding.weight)


class Resampling(nn.Module):
    """tf/functions.py:60-132.  `patch_size` = [from, to]; the token count changes by pool_size = (to / from)^2 (4 for
    every adjacent pair of tf/model.py:12) - down when the patches grow, up when they shrink ('standard' only: 'max' /
    'avg' / 'conv' can only merge)."""

    def __init__(self, img_size: int = 128, patch_size: List[int] = (8, 16), num_channels: int = 1, projection_dim: Optional[int] = 256,
                 resampling_type: str = "standard"):
        super().__init__()
        assert resampling_type in ["max", "avg", "standard", "conv"], "Resampling type must be either 'max', 'avg' or 'standard'."
        self.img_size, self.patch_size, self.num_channels, self.resampling_type = img_size, list(patch_size), num_channels, resampling_type
        self.num_patches = [(img_size // p) ** 2 for p in self.patch_size]
        self.pool_size = self.num_patches[0] // max(self.num_patches[1], 1)
        if resampling_type in ("max", "avg"):
            assert projection_dim is not None, "Projection_dim must be specified when performing 'max' or 'avg' pooling type."
            assert self.pool_size == 4, "token pooling is built for pool_size 4 (adjacent patch sizes a factor 2 apart)"
            self.projection_dim = projection_dim
            self.position_embedding = nn.Embedding(self.num_patches[-1], projection_dim)
        else:
            pd = [projection_dim if projection_dim is not None else num_channels * p ** 2 for p in self.patch_size]
            self.projection_dim = pd
            self.position_embedding = nn.Embedding(self.num_patches[-1], pd[-1])
            if resampling_type == "standard":
                self.linear = Dense(num_channels * self.patch_size[1] ** 2, pd[-1])
            else:
                assert self.pool_size == 4, "'conv' resampling is built for pool_size 4"
                k = self.pool_size // 2
                self.conv = nn.Conv2d(self.num_patches[0], self.num_patches[-1], k, stride=k)      # patches are the channels
                self.linear = Dense(pd[0] // 4, pd[-1])

    def forward(self, encoded):
        C_ = self.num_channels
        if self.resampling_type in ("max", "avg"):
            return _Pool4Fn.apply(encoded, self.position_embedding.weight, 0 if self.resampling_type == "max" else 1)
        if self.resampling_type == "standard":
            x = _RetileFn.apply(encoded, C_, self.img_size, self.patch_size[0], self.patch_size[1])
            return _AddPosFn.apply(self.linear(x), self.position_embedding.weight)
        # 'conv': Conv2D(num_patches[-1], 2, strides 2) over the (s, s) grid of every patch with the PATCHES as channels,
        # as a GEMM over an im2col view (the gather is a view / permute copy, the contraction is vu_gemm)
        B, N0, P0 = encoded.shape
        s = int(round(math.sqrt(P0 // C_)))
        N1 = self.num_patches[-1]
        a = encoded.reshape(B, N0, C_, s // 2, 2, s // 2, 2).permute(0, 2, 3, 5, 1, 4, 6).reshape(B * C_ * (s // 2) ** 2, N0 * 4)
        y = _DenseFn.apply(a, self.conv.weight.reshape(N1, N0 * 4), self.conv.bias)                # (B C s/2 s/2, N1)
        y = y.reshape(B, C_, s // 2, s // 2, N1).permute(0, 4, 1, 2, 3).reshape(B, N1, P0 // 4)
        return _AddPosFn.apply(self.linear(y), self.position_embedding.weight)


class SkipConnection(nn.Module):
    """tf/functions.py:371-395: MultiHeadAttention(num_heads, projection_dim, projection_dim, attn_drop)(q, v)."""

    def __init__(self, projection_dim: int, num_heads: int = 8, attn_drop: float = 0.2):
        super().__init__()
        self.Attn = KerasMultiHeadAttention(projection_dim, num_heads, projection_dim, attn_drop)

    def forward(self, q, v, seed=0, stream_id=0):
        return self.Attn(q, v, seed=seed, stream_id=stream_id)


class HViT_UNet(nn.Module):
    """tf/model.py:9-209 with `original_attn=True` (Keras MHA blocks): PatchEncoder -> [encoder blocks -> Resampling] x (L-1)
    -> bottleneck -> [Resampling -> decoder blocks -> SkipConnection] x (L-1) -> Y = X + unpatch(tokens) (:208).
    `resampling_type` 'standard' keeps projection_dim = C p^2 per level; 'max' / 'avg' need one `projection_dim` for all
    levels - and then the final tokens cannot be un-patched into the image unless projection_dim = C patch_size[0]^2."""

    def __init__(self, img_size: int = 128, patch_size: List[int] = (8, 16, 32), projection_dim: Optional[int] = None, num_channels: int = 3,
                 num_heads: int = 8, transformer_layers: List[int] = (4, 4), size_bottleneck: int = 4, hidden_unit_factor: float = 2.0,
                 drop_attn: float = 0.2, drop_proj: float = 0.2, drop_linear: float = 0.4, resampling_type: str = "standard",
                 original_attn: bool = True, dtype=torch.float32):
        super().__init__()
        patch_size = list(patch_size)
        assert resampling_type in ["max", "avg", "standard"], "Resampling type must be either 'max', 'avg' or 'standard'."
        assert all(img_size % p == 0 for p in patch_size), "Patch sizes must divide image size."
        assert all(patch_size[i] < patch_size[i + 1] for i in range(len(patch_size) - 1)), "Patch sizes must be a strictly increasing sequence."
        if resampling_type in ("max", "avg"):
            # tf/model.py:24-26 builds the decoder's Resampling on the REVERSED patch sizes, where num_patches[0] // num_patches[1]
            # is 0 (tf/functions.py:77): the reference model cannot be assembled with a pooling type either (it fails inside the
            # Keras pooling layer).  Said here, instead of through an assertion about pool sizes further down.
            raise NotImplementedError("HViT_UNet(resampling_type='max' / 'avg'): the decoder would pool with pool_size 0 "
                                      "(tf/functions.py:77 on the reversed patch sizes) - the reference model cannot be built this "
                                      "way either; use 'standard'.  The Resampling LAYER supports 'max' / 'avg' for merging levels.")
        assert (resampling_type in ["max", "avg"] and projection_dim is not None) or (resampling_type == "standard" and projection_dim is None), \
            "If resampling_type is in ['max', 'avg'], projection_dim must be specified. If resampling_type is 'standard', projection_dim is automatically computed."
        if not original_attn:
            raise NotImplementedError("original_attn=False (the Keras ReAttention: an N x N mix over the KEY axis) is not built; "
                                      "the re-attention of this library is the torch model's (vit_unet.torch.model)")
        self.img_size, self.patch_size, self.num_channels, self.compute_dtype = img_size, patch_size, num_channels, dtype
        L = len(patch_size)
        pd = [projection_dim] * L if projection_dim is not None else [num_channels * p ** 2 for p in patch_size]
        hid = [int(hidden_unit_factor * d) for d in pd]
        rev = patch_size[::-1]
        self.projection_dim = pd
        self.PE = PatchEncoder(img_size, patch_size[0], num_channels, pd[0])
        self.Encoder = nn.ModuleList([AttentionTransformerEncoder(num_heads, transformer_layers[i], pd[i], hid[i], drop_attn, drop_proj)
                                      for i in range(L - 1)])
        self.Encoder_RS = nn.ModuleList([Resampling(img_size, patch_size[i:i + 2], num_channels, projection_dim, resampling_type)
                                         for i in range(L - 1)])
        self.BottleNeck = AttentionTransformerEncoder(num_heads, size_bottleneck, pd[-1], hid[-1], drop_attn, drop_proj)
        self.Decoder_RS = nn.ModuleList([Resampling(img_size, rev[i:i + 2], num_channels, projection_dim, resampling_type)
                                         for i in range(L - 1)])
        self.Decoder = nn.ModuleList([AttentionTransformerEncoder(num_heads, transformer_layers[L - (i + 2)], pd[L - (i + 2)], hid[L - (i + 2)],
                                                                  drop_attn, drop_proj) for i in range(L - 1)])
        self.SkipConnections = nn.ModuleList([SkipConnection(pd[L - (i + 2)], num_heads, drop_attn) for i in range(L - 1)])

    def forward(self, X, seed=None):
        B, C_, im, _ = X.shape
        dt = self.compute_dtype
        seed = _next_seed() if (seed is None and self.training) else (seed or 0)
        Xc = X.to(dt).contiguous()
        enc = self.PE(Xc)
        stream, skips = 0, []
        for i in range(len(self.Encoder)):
            enc = self.Encoder[i](enc, seed=seed, stream_id=stream)
            stream += len(self.Encoder[i].Attn)
            skips.append(enc)
            enc = self.Encoder_RS[i](enc)
        enc = self.BottleNeck(enc, seed=seed, stream_id=stream)
        stream += len(self.BottleNeck.Attn)
        skips = skips[::-1]
        for i in range(len(self.Decoder)):
            enc = self.Decoder_RS[i](enc)
            enc = self.Decoder[i](enc, seed=seed, stream_id=stream)
            stream += len(self.Decoder[i].Attn)
            enc = self.SkipConnections[i](skips[i], enc, seed=seed, stream_id=stream)
            stream += 1
        s0 = self.patch_size[0]
        assert enc.shape[-1] == C_ * s0 * s0, "the final tokens must have C * patch_size[0]^2 features to be un-patched (tf/model.py:208)"
        img = _RetileFn.apply(enc, C_, im, s0, im).reshape(B, C_, im, im)
        return _AddFn.apply(Xc, img).float()               # Y = X + unpatch(...)
