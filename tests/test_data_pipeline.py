"""Input pipeline (dataset.py:52-71 + run_denoising.py:52-59).

CPU part: the oracle's restatement of the OpenCV / albumentations arithmetic against properties
and hand values (cv2 and albumentations are absent from the reference tree and this image: parity
with them is unpinned), and the host-side matrix logic of the product.
GPU part: `vu_denoise_prepare` through the C ABI against the oracle - integer / byte work, so the
bar is BIT-EXACT floats."""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vit_unet_oracle as O

DEV = "cuda"


def _batch(B, H, W, C, seed=0):
    rng = np.random.default_rng(seed)
    # smooth + noise so that interpolation errors are visible but bounded
    base = rng.integers(0, 256, (B, H // 8 + 2, W // 8 + 2, C)).astype(np.float32)
    up = F.interpolate(torch.from_numpy(base).permute(0, 3, 1, 2), size=(H, W), mode="bilinear").permute(0, 2, 3, 1).numpy()
    clean = np.clip(up, 0, 255).astype(np.uint8)
    noisy = np.clip(up + rng.normal(0, 25, up.shape), 0, 255).astype(np.uint8)
    return noisy, clean


# ---------------------------------------------------------------- CPU: oracle + host logic
def test_resize_oracle_properties():
    n, _ = _batch(1, 300, 280, 3)
    img = n[0]
    assert np.array_equal(O.resize_u8(img[:224, :224], 224), img[:224, :224])            # same size: copy
    const = np.full((57, 91, 3), 137, np.uint8)
    assert np.array_equal(O.resize_u8(const, 224), np.full((224, 224, 3), 137, np.uint8))   # weights sum to one
    # exact 2x reduction = rounded 2x2 box mean
    a = img[:280, :280]
    box = (a.astype(int).reshape(140, 2, 140, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(O.resize_u8(a, 140), box.astype(np.uint8))
    # within 1 grey level of float bilinear with half-pixel centres (what INTER_LINEAR means)
    ref = F.interpolate(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None], size=(224, 224),
                        mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(O.resize_u8(img, 224).astype(np.float32) - ref).max() <= 1.0
    # hand value: 2 -> 4 upsampling of [0, 200]: centres at -0.25, 0.25, 0.75, 1.25 -> 0, 50, 150, 200
    two = np.array([[[0], [200]], [[0], [200]]], np.uint8)
    assert O.resize_u8(two, 4)[0, :, 0].tolist() == [0, 50, 150, 200]


def test_warp_oracle_properties():
    n, _ = _batch(1, 64, 64, 3, seed=3)
    img = n[0]
    I = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(O.warp_affine_u8(img, I, nearest=False), img)
    assert np.array_equal(O.warp_affine_u8(img, I, nearest=True), img)
    # integer shift by (+5, -3): dst(x,y) = src(x-5, y+3), zeros where the source is outside
    fwd = np.array([[1.0, 0, 5], [0, 1.0, -3]])
    for nearest in (False, True):
        w = O.warp_affine_u8(img, O.invert_affine(fwd), nearest=nearest)
        assert np.array_equal(w[:61, 5:], img[3:, :59])
        assert not w[:, :5].any() and not w[61:].any()
    # half-pixel shift: mean of neighbours, rounded half up
    w = O.warp_affine_u8(img, O.invert_affine(np.array([[1.0, 0, 0.5], [0, 1.0, 0]])), nearest=False)
    exp = (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1
    assert np.array_equal(w[:, 1:], exp.astype(np.uint8))
    # rotation by 90 degrees about the centre maps the image onto its transpose-flip
    M = O.shift_scale_rotate_matrix(64, 90.0, 1.0, 0.0, 0.0)
    w = O.warp_affine_u8(img, O.invert_affine(M), nearest=True)
    assert np.array_equal(w, np.rot90(img, 1))
    # inverse really inverts
    M = O.shift_scale_rotate_matrix(224, 17.0, 1.13, 0.1, -0.05)
    A = np.vstack([M, [0, 0, 1]]) @ np.vstack([O.invert_affine(M), [0, 0, 1]])
    assert np.abs(A - np.eye(3)).max() < 1e-12


def test_normalize_hand_values():
    """run_denoising.py:54 + dataset.py:66: x = ((v/255 - 0.456)/0.224)/255 ; y = v/255."""
    v = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1)
    x, y = O.denoise_prepare(v, v, 16, None)
    assert x.dtype == torch.float32 and x.shape == (1, 1, 16, 16)
    ref = ((np.arange(256) / 255.0 - 0.456) / 0.224) / 255.0
    assert np.abs(x.reshape(-1).numpy() - ref).max() < 1e-8
    assert np.array_equal(y.reshape(-1).numpy(), (np.arange(256) / 255.0).astype(np.float32))


def test_product_matrix_logic_matches_oracle():
    from vit_unet.torch import dataset as D
    fwd = D.shift_scale_rotate_matrices(16, 224, rng=random.Random(7))
    assert fwd.shape == (16, 2, 3)
    sc = np.sqrt(fwd[:, 0, 0] ** 2 + fwd[:, 0, 1] ** 2)
    ang = np.degrees(np.arctan2(fwd[:, 0, 1], fwd[:, 0, 0]))
    assert (sc >= 0.8 - 1e-12).all() and (sc <= 1.2 + 1e-12).all() and (np.abs(ang) <= 20 + 1e-9).all()
    inv = D.invert_affine(fwd)
    for b in range(16):
        assert np.array_equal(inv[b], O.invert_affine(fwd[b]))
    r = random.Random(3)
    a, s, dx, dy = r.uniform(-20, 20), r.uniform(0.8, 1.2), r.uniform(-0.2, 0.2), r.uniform(-0.2, 0.2)
    assert np.array_equal(D.shift_scale_rotate_matrices(1, 224, rng=random.Random(3))[0],
                          O.shift_scale_rotate_matrix(224, a, s, dx, dy))
    assert D.ImageFitter is not None      # the reference keeps ImageFitter in dataset.py (:76)
    with pytest.raises(Exception):        # CPU tensors: the product refuses, no fallback
        D.DenoisingBatchTransform(32, device="cpu")(np.zeros((1, 32, 32, 3), np.uint8), np.zeros((1, 32, 32, 3), np.uint8))


# ---------------------------------------------------------------- GPU: bit-exact against the oracle
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,C,im", [(3, 300, 280, 3, 224), (2, 224, 224, 3, 224), (2, 448, 448, 3, 224),
                                        (2, 100, 131, 1, 64), (1, 512, 512, 1, 512), (5, 37, 53, 3, 32)])
@pytest.mark.parametrize("train", [False, True])
def test_denoise_prepare_bit_exact(B, H, W, C, im, train):
    from vit_unet.torch import dataset as D
    noisy, clean = _batch(B, H, W, C, seed=B + H)
    fwd = D.shift_scale_rotate_matrices(B, im, rng=random.Random(H + W)) if train else None
    t = D.DenoisingBatchTransform(im, train=train)
    got = t(noisy, clean, matrices=fwd)
    rx, ry = O.denoise_prepare(noisy, clean, im, fwd)
    assert got["x"].shape == (B, C, im, im) and got["x"].dtype == torch.float32 and got["x"].is_cuda
    assert torch.equal(got["x"].cpu(), rx), (got["x"].cpu() - rx).abs().max()
    assert torch.equal(got["y"].cpu(), ry), (got["y"].cpu() - ry).abs().max()


@pytest.mark.gpu
def test_denoise_prepare_extreme_affine_and_errors():
    from vit_unet.torch import _lib, dataset as D
    noisy, clean = _batch(2, 64, 64, 3, seed=9)
    # a warp that pushes most of the image out of the frame, and a strong zoom
    fwd = np.stack([O.shift_scale_rotate_matrix(64, 20.0, 0.8, 0.95, -0.95), O.shift_scale_rotate_matrix(64, -20.0, 3.0, 0.0, 0.0)])
    got = D.DenoisingBatchTransform(64, train=True)(noisy, clean, matrices=fwd)
    rx, ry = O.denoise_prepare(noisy, clean, 64, fwd)
    assert torch.equal(got["x"].cpu(), rx) and torch.equal(got["y"].cpu(), ry)
    # draws its own matrices when none are given; output feeds the model's input contract
    out = D.DenoisingBatchTransform(32, train=True, seed=1)(noisy, clean)
    assert out["x"].shape == (2, 3, 32, 32) and torch.isfinite(out["x"]).all()
    L = _lib.lib()
    z = torch.zeros(16, dtype=torch.uint8, device=DEV)
    f = torch.zeros(16, device=DEV)
    st = _lib.stream_ptr()
    assert L.vu_denoise_prepare(_lib.ptr(z), _lib.ptr(z), _lib.ptr(f), _lib.ptr(f), None, 0, None, 1, 2, 2, 2, 2, 0.456, 0.224, st) < 0  # channels
    assert L.vu_denoise_prepare(_lib.ptr(z), _lib.ptr(z), _lib.ptr(f), _lib.ptr(f), None, 0, None, 1, 4, 4, 1, 2, 0.456, 0.224, st) < 0  # scratch
    assert L.vu_denoise_prepare(_lib.ptr(z), _lib.ptr(z), _lib.ptr(f), _lib.ptr(f), None, 0, None, 0, 2, 2, 1, 2, 0.456, 0.224, st) < 0  # empty


# ---------------------------------------------------------------- K-fold driver (run_denoising.py:16-122)
def test_kfold_indices_partition():
    from vit_unet.torch.run import kfold_indices
    seen = []
    for tr, te in kfold_indices(11, 3, seed=1):
        assert len(set(tr) & set(te)) == 0 and len(tr) + len(te) == 11
        assert sorted(te) == list(te)
        seen.extend(te.tolist())
    assert sorted(seen) == list(range(11))                     # every sample is tested exactly once
    assert [len(te) for _, te in kfold_indices(11, 3, seed=1)] == [4, 4, 3]   # sklearn's fold sizes
    with pytest.raises(AssertionError):
        list(kfold_indices(3, 5))


@pytest.mark.gpu
def test_run_denoising_kfold_end_to_end(tmp_path):
    """Two folds of the Lite preset on a handful of synthetic uint8 pairs: decode-free loader ->
    device pipeline -> fused HIP train step -> checkpoints -> reload best -> per-image PSNR."""
    from vit_unet.torch.run import run_denoising
    noisy, clean = _batch(6, 96, 80, 3, seed=21)
    seen = []
    res = run_denoising(noisy, clean, n_epochs=2, folds=2, model_string="lite", lr=1e-4, batch_size=2, im_size=224,
                        folder=str(tmp_path / "models"), seed=3, callbacks=[lambda fold, log: seen.append((fold, log["epoch"]))])
    assert [len(p) for p in res["psnr"]] == [3, 3]
    assert all(np.isfinite(p).all() for p in res["psnr"]) and np.isfinite(res["psnr_mean"])
    assert seen == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert all("train" in h and "val" in h for hist in res["history"] for h in hist)
    assert (tmp_path / "models" / "best-checkpoint.bin").exists() and (tmp_path / "models" / "last-checkpoint.bin").exists()
    with pytest.raises(AssertionError):
        run_denoising(noisy, clean[:5])


@pytest.mark.gpu
def test_image_fitter_fused_dice(tmp_path):
    """ImageFitter with functions.DiceLoss() takes the fused HIP step (TrainStep(loss='dice'))."""
    from vit_unet.torch import functions as Fn, model as M
    from vit_unet.torch.fitter import ImageFitter
    m = M.HViT_UNet(depth=1, depth_te=1, size_bottleneck=1, preprocessing="conv", im_size=64, patch_size=16,
                    num_channels=1, hidden_dim=16, num_heads=2, attn_drop=0.0, proj_drop=0.0, linear_drop=0.0).to(DEV)
    g = torch.Generator().manual_seed(0)
    data = [{"x": torch.rand(4, 1, 64, 64, generator=g), "y": (torch.rand(4, 1, 64, 64, generator=g) < 0.1).float()}] * 3
    f = ImageFitter(m, loss=Fn.DiceLoss(), device=DEV, folder=str(tmp_path), lr=1e-3)
    hist = f.fit(data, data, n_epochs=3)
    assert f._fused is not None and f._fused.loss_kind == "dice"
    assert hist[-1]["train"] < hist[0]["train"] and np.isfinite(hist[-1]["val"])
    g2 = ImageFitter(m, loss=Fn.DiceLoss(apply_sigmoid=False), device=DEV, folder=str(tmp_path))
    assert g2._fused_kind() is None        # plain Dice on raw outputs: autograd path
