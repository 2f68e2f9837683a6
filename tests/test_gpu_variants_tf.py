"""SURVEY section 8 row f4, second half: the Keras-only layers (vit_unet/torch/variants_tf.py on csrc/vu_tfops.hip + vu_gemm)
against the oracle's restatement of the reference text (oracle tf_* functions; /root/reference/vit_unet/tf/functions.py:60-132,
:135-182, :258-311, :371-395, tf/model.py:188-209).  TensorFlow is not in this image and the reference holds no outputs of these
layers: parity is against the oracle only ("unpinned").  fp32 storage: 2e-5 forward / 2e-4 backward per op,
5e-4 / 5e-3 for chains; bf16 storage: 3e-2 / 5e-2 with the oracle following the same rounding points."""
import numpy as np
import pytest
import torch

import vit_unet_oracle as O
from vit_unet.torch import _lib
from vit_unet.torch import variants_tf as T
from vit_unet.torch._lib import check, lib, ptr, stream_ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def serr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def _perturb(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif "position_embedding" in k:
                p.mul_(0.3)
    return m


def _oracle_params(m):
    p = {k: v.detach().cpu().float().clone() for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    for k, _ in m.named_parameters():
        p[k].requires_grad_(True)
    return p


def _check_grads(m, p, tol, skip=()):
    for k, q in m.named_parameters():
        if any(s in k for s in skip):
            continue
        assert q.grad is not None, k
        if k.endswith("key.bias"):
            # analytically zero: a bias on the keys shifts every score of a row by the same q . b, which the softmax ignores;
            # what both sides hold is rounding residue - held to the level of the query-bias gradient's rounding
            ref = p[k.replace("key.bias", "query.bias")].grad.abs().max().item()
            assert q.grad.abs().max().item() < max(tol, 1e-3) * ref + 1e-6, (k, q.grad.abs().max().item(), ref)
            continue
        assert serr(q.grad, p[k].grad) < tol, (k, serr(q.grad, p[k].grad))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gelu_dropout_add(dt):
    g = torch.Generator().manual_seed(2)
    x = (2 * torch.randn(3, 50, 64, generator=g)).to(dt)
    dy = torch.randn(3, 50, 64, generator=g).to(dt)
    xd = x.to(DEV).requires_grad_(True)
    y = T._GeluFn.apply(xd)
    y.backward(dy.to(DEV))
    xr = x.float().clone().requires_grad_(True)
    yr = torch.nn.functional.gelu(xr)
    yr.backward(dy.float())
    tol = 2e-6 if dt == torch.float32 else 8e-3
    assert serr(y, yr) < tol and serr(xd.grad, xr.grad) < tol
    # dropout: the oracle's mask replay, element for element; the backward applies the same mask
    xd2 = x.to(DEV).requires_grad_(True)
    z = T._dropout(xd2, 0.3, True, 11, 5)
    z.backward(dy.to(DEV))
    keep = O.keep_mask(x.numel(), 0.3, 11, 5).reshape(x.shape).float()
    etol = 2e-7 if dt == torch.float32 else 4e-3          # (x * (1 / 0.7) in the kernel, x / 0.7 here: one rounding apart)
    assert torch.equal(z.detach().float().cpu() != 0, (x.float() * keep) != 0)      # the mask itself: element for element
    assert serr(z, x.float() * keep / 0.7) < etol
    assert serr(xd2.grad, dy.float() * keep / 0.7) < etol
    a, b = x.to(DEV), dy.to(DEV)
    assert torch.equal(T._AddFn.apply(a, b).float().cpu(), (x.float() + dy.float()).to(dt).float())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["max", "avg"])
@pytest.mark.parametrize("B,N,P", [(2, 64, 48), (1, 256, 192), (3, 16, 12)])
def test_token_pool4(dt, mode, B, N, P):
    g = torch.Generator().manual_seed(N + P)
    x = torch.randn(B, N, P, generator=g).to(dt)
    pos = torch.randn(N // 4, P, generator=g)
    dy = torch.randn(B, N // 4, P, generator=g).to(dt)
    xd, pd = x.to(DEV).requires_grad_(True), pos.to(DEV).requires_grad_(True)
    y = T._Pool4Fn.apply(xd, pd, 0 if mode == "max" else 1)
    y.backward(dy.to(DEV))
    xr, pr = x.float().clone().requires_grad_(True), pos.clone().requires_grad_(True)
    yr = O.tf_token_pool4(xr, mode) + pr
    yr.backward(dy.float())
    tol = 1e-6 if dt == torch.float32 else 8e-3
    assert y.shape == (B, N // 4, P)
    assert serr(y, yr) < tol
    if mode == "max":
        # ties inside a pooling window (frequent in bf16: 8 significant bits): torch splits the gradient between the tied
        # maxima, TensorFlow - and the kernel - route it to one of them; compare where the maximum is unique
        xf = x.float()
        N8 = N // 8
        src = torch.tensor([8 * (m % N8) + 2 * (m // N8) for m in range(N // 4)])
        win = torch.stack([xf[:, src + o] for o in (0, 1, 4, 5)])                       # (4, B, N/4, P)
        uniq = (win == win.max(0).values).sum(0) == 1                                  # (B, N/4, P)
        keep = torch.zeros(B, N, P, dtype=torch.bool)
        for o in (0, 1, 4, 5):
            keep[:, src + o] = uniq
        assert uniq.float().mean() > 0.9
        assert serr(xd.grad.cpu().float() * keep, xr.grad * keep) < tol
        # and every window's gradient adds up to dy whichever element received it
        gs = sum(xd.grad.cpu().float()[:, src + o] for o in (0, 1, 4, 5))
        assert serr(gs, dy.float()) < tol
    else:
        assert serr(xd.grad, xr.grad) < tol
    assert serr(pd.grad, pr.grad) < (1e-5 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,ps,pd", [("max", [4, 8], 48), ("avg", [4, 8], 48), ("standard", [4, 8], None), ("standard", [8, 4], None),
                                        ("conv", [4, 8], None)])
def test_resampling_layer(dt, kind, ps, pd):
    im, C = 32, 3
    m = _perturb(T.Resampling(im, ps, C, pd, kind), 3).to(DEV)
    p = _oracle_params(m)
    N0 = (im // ps[0]) ** 2
    P0 = pd if pd is not None else C * ps[0] ** 2
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, N0, P0, generator=g).to(dt)
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    dy = torch.randn(y.shape, generator=g).to(dt)
    y.backward(dy.to(DEV))
    xr = x.float().clone().requires_grad_(True)
    st = dt if dt == torch.bfloat16 else None
    yr = O.tf_resampling(xr, p, "", kind=kind, img_size=im, patch_size=ps, C=C, storage=st)
    yr.backward(dy.float())
    ft, bt = (2e-5, 2e-4) if dt == torch.float32 else (3e-2, 5e-2)
    assert y.shape == yr.shape
    assert serr(y, yr) < ft
    if not (kind == "max" and dt == torch.bfloat16):       # (bf16 ties in a max window: test_token_pool4)
        assert serr(xd.grad, xr.grad) < bt
    _check_grads(m, p, bt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_keras_mha_cross(dt, drop):
    """SkipConnection (tf/functions.py:371-395): query from the encoder, key = value from the decoder, odd key count (ld padding)."""
    D, H, Nq, Nk = 48, 4, 20, 36
    m = _perturb(T.SkipConnection(D, H, drop), 5).to(DEV).train()
    p = _oracle_params(m)
    g = torch.Generator().manual_seed(6)
    q, v = torch.randn(2, Nq, D, generator=g).to(dt), torch.randn(2, Nk, D, generator=g).to(dt)
    dy = torch.randn(2, Nq, D, generator=g).to(dt)
    qd, vd = q.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    y = m(qd, vd, seed=9, stream_id=2)
    y.backward(dy.to(DEV))
    qr, vr = q.float().clone().requires_grad_(True), v.float().clone().requires_grad_(True)
    yr = O.keras_mha(qr, vr, p, "Attn.", num_heads=H, dropout=drop, training=True, seed=9, stream=2, storage=dt if dt == torch.bfloat16 else None)
    yr.backward(dy.float())
    ft, bt = (2e-5, 2e-4) if dt == torch.float32 else (3e-2, 5e-2)
    assert serr(y, yr) < ft
    assert serr(qd.grad, qr.grad) < bt and serr(vd.grad, vr.grad) < bt
    _check_grads(m, p, bt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_attention_transformer_encoder(dt):
    D, H, N, hid = 48, 4, 64, 96
    m = _perturb(T.AttentionTransformerEncoder(H, 2, D, hid, 0.2, 0.1), 7).to(DEV).train()
    p = _oracle_params(m)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, N, D, generator=g).to(dt)
    dy = torch.randn(2, N, D, generator=g).to(dt)
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd, seed=21, stream_id=3)
    y.backward(dy.to(DEV))
    xr = x.float().clone().requires_grad_(True)
    yr = O.tf_attention_te(xr, p, "", layers=2, num_heads=H, attn_drop=0.2, proj_drop=0.1, training=True, seed=21, stream=3,
                           storage=dt if dt == torch.bfloat16 else None)
    yr.backward(dy.float())
    ft, bt = (5e-5, 5e-4) if dt == torch.float32 else (3e-2, 6e-2)
    assert serr(y, yr) < ft
    assert serr(xd.grad, xr.grad) < bt
    _check_grads(m, p, bt)


@pytest.mark.parametrize("kind,pd", [("standard", None)])
@pytest.mark.parametrize("training", [False, True])
def test_tf_hvit_unet_model(kind, pd, training):
    """The whole Keras-variant model (tf/model.py:9-209, original_attn=True) incl. the input residual, fp32, dropout replayed.
    ('max' / 'avg' cannot be assembled into the U: the decoder's Resampling would need pool_size = num_patches[0] //
    num_patches[1] = 0, tf/functions.py:77 - in the reference as here; those modes are tested as layers.)"""
    kw = dict(img_size=32, patch_size=[4, 8, 16], projection_dim=pd, num_channels=3, num_heads=2, transformer_layers=[1, 2], size_bottleneck=1,
              hidden_unit_factor=2.0, drop_attn=0.2, drop_proj=0.1, resampling_type=kind)
    m = _perturb(T.HViT_UNet(**kw), 11).to(DEV).train(training)
    p = _oracle_params(m)
    g = torch.Generator().manual_seed(12)
    X = torch.rand(2, 3, 32, 32, generator=g)
    dY = torch.randn(2, 3, 32, 32, generator=g)
    Xd = X.to(DEV).requires_grad_(True)
    Y = m(Xd, seed=31)
    Y.backward(dY.to(DEV))
    Xr = X.clone().requires_grad_(True)
    Yr = O.tf_forward(p, Xr, img_size=32, patch_size=[4, 8, 16], num_channels=3, num_heads=2, transformer_layers=[1, 2], size_bottleneck=1,
                      drop_attn=0.2, drop_proj=0.1, resampling_type=kind, training=training, seed=31)
    Yr.backward(dY)
    assert Y.shape == X.shape
    assert serr(Y, Yr) < 5e-4
    assert serr(Xd.grad, Xr.grad) < 5e-3
    _check_grads(m, p, 5e-3)
    # the input residual is really there (tf/model.py:208): the output moves one for one with the input at fixed tokens
    assert (Y - Xd).abs().max().item() > 1e-3


def test_tf_model_rejects_keras_reattention():
    with pytest.raises(NotImplementedError, match="pool_size 0"):       # 'avg' in the U: the decoder direction has pool_size 0
        T.HViT_UNet(img_size=32, patch_size=[4, 8], projection_dim=48, num_channels=3, num_heads=2, transformer_layers=[1], size_bottleneck=1,
                    resampling_type="avg")
    with pytest.raises(NotImplementedError):
        T.HViT_UNet(img_size=32, patch_size=[4, 8], num_channels=3, num_heads=2, transformer_layers=[1], size_bottleneck=1, original_attn=False)
