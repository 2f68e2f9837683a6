"""the notebook variants (vit_unet/torch/variants.py) against the oracle's restatement of the
notebook text (oracle.fft2_real / fformer_block / notebook_te_block / notebook_forward).  fp32 storage: 2e-4 forward,
5e-3 backward (the tolerances of the tiny-model tests: the eval-free train-mode chains amplify fp32 noise by
1/sqrt(var) of the re-attention BatchNorm); bf16 storage: 6e-2 on a block.  Parity is against the oracle only - the
notebook cannot be run as committed."""
import numpy as np
import pytest
import torch

import vit_unet_oracle as O
from vit_unet.torch import variants as V

pytestmark = pytest.mark.gpu
DEV = "cuda"


def serr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def _perturb(m, seed):
    """default inits leave LayerNorm / BatchNorm at identity and the embedding N(0,1): move everything a little so that
    every parameter matters, deterministically"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.dim() == 1 or k.endswith("LN.weight") or k.endswith("LN.bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif "position_embedding" in k:
                p.mul_(0.3)
    return m


def _oracle_params(m):
    p = {k: v.detach().cpu().float().clone() for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    for k, _ in m.named_parameters():
        p[k].requires_grad_(True)
    return p


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 49, 192), (3, 196, 48), (2, 3, 64, 64), (1, 784, 12), (2, 50, 36)])
def test_fft2_real_forward_backward(dt, shape):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape, generator=g).to(dt)
    dy = torch.randn(shape, generator=g).to(dt)
    xd = x.to(DEV).requires_grad_(True)
    y = V.fft2_real(xd)
    y.backward(dy.to(DEV))
    xr = x.float().requires_grad_(True)
    yr = O.fft2_real(xr)
    yr.backward(dy.float())
    tol = 2e-5 if dt == torch.float32 else 3e-2
    assert y.dtype == dt and y.shape == x.shape
    assert serr(y, yr) < tol
    assert serr(xd.grad, xr.grad) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["eval", "train_drop"])
def test_fformer_encoder_block(dt, mode):
    N, D, hid = 196, 48, 32
    torch.manual_seed(3)
    m = _perturb(V.FformerEncoder(N, D, hid, dropout=0.2 if mode == "train_drop" else 0.0), 4)
    m.train(mode != "eval")
    p = _oracle_params(m)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, N, D, generator=g).to(dt)
    dy = torch.randn(2, N, D, generator=g).to(dt)
    md = m.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    y = md(xd, seed=77, stream_id=5)
    y.backward(dy.to(DEV))
    xr = x.float().requires_grad_(True)
    yr = O.fformer_block(xr, p, "", training=(mode != "eval"), linear_drop=0.2 if mode == "train_drop" else 0.0, seed=77,
                         stream=5, storage=(dt if dt != torch.float32 else None))
    yr.backward(dy.float())
    ft, bt = (2e-4, 2e-3) if dt == torch.float32 else (6e-2, 6e-2)
    assert serr(y, yr) < ft
    assert serr(xd.grad, xr.grad) < bt
    for k, q in md.named_parameters():
        assert serr(q.grad, p[k].grad) < bt, k     # LN.weight / LN.bias: the sum of both uses


@pytest.mark.parametrize("mode", ["train", "train_drop"])
def test_notebook_transformer_block_single_layernorm_1x1_qkv(mode):
    N, C_, s, H, hid = 49, 3, 8, 4, 32
    D = C_ * s * s
    drop = 0.2 if mode == "train_drop" else 0.0
    torch.manual_seed(5)
    m = _perturb(V.NotebookTransformerEncoder(N, D, hid, H, drop, drop, 0.0, num_channels=C_), 6).train()
    assert m.ReAttn.qconv2d.weight.shape == (C_, C_, 1, 1)
    p = _oracle_params(m)
    cfg = O.Config(depth=1, depth_te=1, size_bottleneck=1, preprocessing="none", im_size=7 * 16, patch_size=16, num_channels=C_,
                   hidden_dim=hid, num_heads=H, attn_drop=drop, proj_drop=drop)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, N, D, generator=g)
    dy = torch.randn(2, N, D, generator=g)
    md = m.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    y = md(xd, seed=99, stream_id=2)
    y.backward(dy.to(DEV))
    xr = x.clone().requires_grad_(True)
    yr = O.notebook_te_block(xr, p, "", cfg, training=True, seed=99, stream=2)
    yr.backward(dy)
    assert serr(y, yr) < 2e-4
    assert serr(xd.grad, xr.grad) < 5e-3
    for k, q in md.named_parameters():
        if k.endswith("reatten_matrix.bias"):
            continue
        assert q.grad.shape == p[k].grad.shape
        assert serr(q.grad, p[k].grad) < 5e-3, k


@pytest.mark.parametrize("block,pre,depth,depth_te", [("reattention", "conv", 2, 2), ("reattention", "none", 1, 1),
                                                     ("fformer", "conv", 2, 2), ("fformer", "fourier_in", 1, 1)])
def test_notebook_model_forward_backward(block, pre, depth, depth_te):
    """the whole notebook model, fp32, train mode with dropout: output, input gradient and every parameter gradient.
    The notebook indexes its skip modules with (i - 1) // depth_te: right for depth_te >= 2, and for depth_te = 1 only
    when there is a single level (index -1 = the only module; with more levels the notebook's own shapes disagree).
    ('fourier_in': only the PatchEncoder's Re(fft2) input branch, with the output branch of 'none' - the notebook's own
    'fourier' output is the D5 bug.)"""
    nb, C_, H = 1, 3, 4
    im, ps, hid = 64, 16, 16
    npatch = (im // ps) ** 2
    torch.manual_seed(11)
    m = V.NotebookViT_UNet(depth, depth_te, nb, "none" if pre == "fourier_in" else pre, npatch, ps, C_ * ps * ps, hid, H,
                           0.2, 0.2, 0.0, block=block)
    if pre == "fourier_in":
        m.PE.preprocessing = "fourier"
    _perturb(m, 12).train()
    p = _oracle_params(m)
    cfg = O.Config(depth=depth, depth_te=depth_te, size_bottleneck=nb, preprocessing="none" if pre == "fourier_in" else pre,
                   im_size=im, patch_size=ps, num_channels=C_, hidden_dim=hid, num_heads=H, attn_drop=0.2, proj_drop=0.2)
    g = torch.Generator().manual_seed(13)
    X = torch.rand(2, C_, im, im, generator=g)
    dY = torch.randn(2, C_, im, im, generator=g)
    md = m.to(DEV)
    Xd = X.to(DEV).requires_grad_(True)
    Y = md(Xd, seed=4242)
    Y.backward(dY.to(DEV))
    Xr = X.clone().requires_grad_(True)
    if pre == "fourier_in":
        # the input transform is applied in front of the 'none' model, whose PatchEncoder is then tokens + embedding
        Yr = O.notebook_forward(p, cfg, O.fft2_real(Xr), training=True, seed=4242, block=block)
    else:
        Yr = O.notebook_forward(p, cfg, Xr, training=True, seed=4242, block=block)
    Yr.backward(dY)
    assert Y.shape == X.shape and Y.dtype == torch.float32
    if block == "reattention" and depth == 2:
        # Eleven re-attention blocks in a row: every train-mode BatchNorm over a near-constant map amplifies fp32
        # summation-order noise by 1/sqrt(var), and the chain is ill-conditioned IN THE ORACLE ITSELF (its fp32 and fp64
        # runs differ by 8e-4 forward and up to 3 % on a gradient).  The device run is one more fp32 realisation: it is
        # held to a small multiple of the oracle's own fp32-to-fp64 distance, tensor by tensor.
        p64 = {k: v.detach().double().clone() for k, v in p.items()}
        for k, _ in md.named_parameters():
            p64[k].requires_grad_(True)
        X64 = X.double().requires_grad_(True)
        Y64 = O.notebook_forward(p64, cfg, X64, training=True, seed=4242, block=block)
        Y64.backward(dY.double())

        def bound(r32, r64, floor):
            return 10.0 * serr(r32, r64) + floor
        assert serr(Y, Y64) < bound(Yr, Y64, 2e-4)
        assert serr(Xd.grad, X64.grad) < bound(Xr.grad, X64.grad, 5e-3)
        for k, q in md.named_parameters():
            if k.endswith("reatten_matrix.bias"):
                continue
            assert serr(q.grad, p64[k].grad) < bound(p[k].grad, p64[k].grad, 5e-3), k
        return
    assert serr(Y, Yr) < 2e-4
    assert serr(Xd.grad, Xr.grad) < 5e-3
    for k, q in md.named_parameters():
        if k.endswith("reatten_matrix.bias"):
            continue
        assert q.grad is not None, k
        assert serr(q.grad, p[k].grad) < 5e-3, k
