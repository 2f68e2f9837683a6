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


# ---------------------------------------------------------------- SegmentationDataset pipeline (dataset.py:9-41)
def _seg_batch(B, H, W, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.integers(-1200, 2500, (B, 1, H // 8 + 2, W // 8 + 2)).astype(np.float32)
    up = F.interpolate(torch.from_numpy(base), size=(H, W), mode="bilinear")[:, 0].numpy()
    img = np.clip(up + rng.normal(0, 40, up.shape), -32768, 32767).astype(np.int16)
    mask = (up > 600).astype(np.uint8) + (up > 1800).astype(np.uint8)         # labels 0 / 1 / 2
    return img, mask


def test_seg_oracle_properties():
    img, mask = _seg_batch(1, 96, 80, seed=1)
    i0, m0 = img[0], mask[0]
    ri, rm = O.seg_resize(i0, m0, 96, 80)
    assert np.array_equal(ri, i0) and np.array_equal(rm, m0)                     # same size: copy
    ri, rm = O.seg_resize(np.full((57, 91), -700, np.int16), np.full((57, 91), 2, np.uint8), 128, 128)
    assert (ri == -700).all() and (rm == 2).all()                                # weights sum to one
    ri, rm = O.seg_resize(i0, m0, 48, 40)                                         # exact 2x: box mean / top-left label
    assert np.array_equal(ri, (i0.astype(int).reshape(48, 2, 40, 2).sum(axis=(1, 3)) + 2) >> 2)
    assert np.array_equal(rm, m0[::2, ::2])
    ref = F.interpolate(torch.from_numpy(i0.astype(np.float32))[None, None], size=(128, 128), mode="bilinear")[0, 0].numpy()
    assert np.abs(O.seg_resize(i0, None, 128, 128)[0] - ref).max() <= 0.5 + 1e-3  # float bilinear, rounded
    # warp: identity, integer shift, half-pixel mean (round half even on a float sum)
    I = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    wi, wm = O.seg_warp(i0.astype(np.int64), m0, I)
    assert np.array_equal(wi, i0) and np.array_equal(wm, m0)
    wi, wm = O.seg_warp(i0.astype(np.int64), m0, O.invert_affine(np.array([[1.0, 0, 5], [0, 1.0, -3]])))
    assert np.array_equal(wi[:93, 5:], i0[3:, :75]) and not wi[:, :5].any() and not wm[93:].any()
    wi, _ = O.seg_warp(i0.astype(np.int64), None, O.invert_affine(np.array([[1.0, 0, 0.5], [0, 1.0, 0]])))
    assert np.array_equal(wi[:, 1:], np.rint((i0[:, :-1].astype(np.float64) + i0[:, 1:]) / 2).astype(np.int64))
    # window + label smoothing hand values
    v = np.array([[[-2000, -1024, 0, 1024, 3000]]], np.int16)
    x, y = O.seg_prepare(v, np.array([[[0, 1, 1, 0, 1]]], np.uint8), (1, 5), None, lo=-1024, hi=1024, ls=0.1)
    assert x.shape == (1, 1, 1, 5) and x.reshape(-1).tolist() == [0.0, 0.0, 0.5, 1.0, 1.0]
    assert np.allclose(y.reshape(-1).numpy(), [0.05, 0.95, 0.95, 0.05, 0.95], atol=1e-7)
    x, y = O.seg_prepare(v, None, (1, 5))
    assert y is None and x is not None


def test_dataset_classes_host_side(tmp_path):
    """Same names and constructor arguments as dataset.py:9-17 / :44-51; items are the decoded pairs."""
    from PIL import Image
    from vit_unet.torch import dataset as D
    rng = np.random.default_rng(0)
    (tmp_path / "c").mkdir(), (tmp_path / "n").mkdir()
    rgb = rng.integers(0, 256, (20, 24, 3), dtype=np.uint8)
    for sub in "cn":
        Image.fromarray(rgb).save(tmp_path / sub / "a.png")
    ds = D.DenoisingDataset(["a"], augments=None, clean_folder=str(tmp_path / "c"), noisy_folder=str(tmp_path / "n"), im_size=16)
    assert len(ds) == 1
    it = ds[0]
    assert set(it) == {"x", "y"} and it["x"].dtype == np.uint8 and np.array_equal(it["x"], rgb[:, :, ::-1])    # BGR like cv2.imread
    with pytest.raises(TypeError):
        D.DenoisingDataset(["a"], augments=lambda **k: k)
    import pandas as pd
    df = pd.DataFrame({"image": ["i0", "i1"], "mask": ["m0", "m1"], "mask_index": [0, 1]})
    vol = {"m0": rng.integers(0, 2, (12, 10, 2)), "m1": rng.integers(0, 2, (12, 10, 2))}
    sl = {"i0": rng.integers(-1000, 1000, (12, 10)), "i1": rng.integers(-1000, 1000, (12, 10))}
    sd = D.SegmentationDataset(df, augments=None, is_test=False, data_folder="output", im_size=(8, 8), ls=0.1,
                               read_image=lambda p: sl[p].astype(np.int16), read_mask=lambda p, k: vol[p][:, :, k].astype(np.uint8))
    assert len(sd) == 2 and np.array_equal(sd[1]["y"], vol["m1"][:, :, 1]) and sd[1]["x"].dtype == np.int16
    assert "y" not in D.SegmentationDataset(df, is_test=True, read_image=lambda p: sl[p].astype(np.int16))[0]
    fwd = D.shift_scale_rotate_matrices(1, (8, 12), rng=random.Random(3))[0]     # non-square: centre (w/2-.5, h/2-.5)
    r = random.Random(3)
    a, s, dx, dy = r.uniform(-20, 20), r.uniform(0.8, 1.2), r.uniform(-0.2, 0.2), r.uniform(-0.2, 0.2)
    ca, sa = s * np.cos(np.radians(a)), s * np.sin(np.radians(a))
    assert np.allclose(fwd, [[ca, sa, (1 - ca) * 5.5 - sa * 3.5 + dx * 12], [-sa, ca, sa * 5.5 + (1 - ca) * 3.5 + dy * 8]])


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,oh,ow", [(3, 300, 280, 128, 128), (2, 128, 128, 128, 128), (2, 512, 512, 256, 256),
                                         (2, 100, 131, 64, 96), (1, 512, 512, 512, 512), (5, 37, 53, 32, 48)])
@pytest.mark.parametrize("train", [False, True])
def test_seg_prepare_bit_exact(B, H, W, oh, ow, train):
    from vit_unet.torch import dataset as D
    img, mask = _seg_batch(B, H, W, seed=B + H)
    fwd = D.shift_scale_rotate_matrices(B, (oh, ow), rng=random.Random(H + W)) if train else None
    t = D.SegmentationBatchTransform((oh, ow), train=train, window=(-1000.0, 2000.0), ls=0.1)
    got = t(img, mask, matrices=fwd)
    rx, ry = O.seg_prepare(img, mask, (oh, ow), fwd, lo=-1000.0, hi=2000.0, ls=0.1)
    assert got["x"].shape == (B, 1, oh, ow) and got["x"].dtype == torch.float32 and got["x"].is_cuda
    assert torch.equal(got["x"].cpu(), rx), (got["x"].cpu() - rx).abs().max()
    assert torch.equal(got["y"].cpu(), ry), (got["y"].cpu() - ry).abs().max()


@pytest.mark.gpu
def test_seg_dataset_to_device_batches_and_errors():
    from vit_unet.torch import _lib, dataset as D
    import pandas as pd
    img, mask = _seg_batch(5, 72, 64, seed=4)
    big_i, big_m = _seg_batch(1, 90, 100, seed=5)            # one item of another size: grouped, order kept
    imgs, masks = list(img) + [big_i[0]], list(mask) + [big_m[0]]
    df = pd.DataFrame({"image": range(6), "mask": range(6), "mask_index": [0] * 6})
    ds = D.SegmentationDataset(df, im_size=(32, 32), ls=0.0, read_image=lambda p: imgs[p], read_mask=lambda p, k: masks[p])
    batches = list(D.DeviceBatches(ds, batch_size=4))
    assert [b["x"].shape[0] for b in batches] == [4, 2] and batches[0]["x"].shape[1:] == (1, 32, 32)
    rx, ry = O.seg_prepare(img[4:5], mask[4:5], (32, 32))
    assert torch.equal(batches[1]["x"][0:1].cpu(), rx) and torch.equal(batches[1]["y"][0:1].cpu(), ry)
    rx, ry = O.seg_prepare(big_i, big_m, (32, 32))
    assert torch.equal(batches[1]["x"][1:2].cpu(), rx) and torch.equal(batches[1]["y"][1:2].cpu(), ry)
    test_only = D.SegmentationDataset(df, is_test=True, im_size=(32, 32), read_image=lambda p: imgs[p])
    out = test_only.transform([test_only[0], test_only[1]])
    assert set(out) == {"x"} and out["x"].shape == (2, 1, 32, 32)
    L = _lib.lib()
    z16 = torch.zeros(16, dtype=torch.int16, device=DEV)
    z8 = torch.zeros(16, dtype=torch.uint8, device=DEV)
    f = torch.zeros(16, device=DEV)
    st = _lib.stream_ptr()
    args = lambda **k: (_lib.ptr(z16), _lib.ptr(z8), _lib.ptr(f), _lib.ptr(f), None, 0, None, k.get("B", 1), 4, 4, k.get("oh", 4), 4,
                        k.get("lo", 0.0), k.get("hi", 1.0), k.get("ls", 0.0), st)
    assert L.vu_seg_prepare(*args()) == 0
    assert L.vu_seg_prepare(*args(hi=0.0)) < 0 and L.vu_seg_prepare(*args(ls=1.0)) < 0
    assert L.vu_seg_prepare(*args(oh=2)) < 0 and L.vu_seg_prepare(*args(B=0)) < 0      # scratch missing / empty


@pytest.mark.gpu
def test_denoising_dataset_device_batches(tmp_path):
    from PIL import Image
    from vit_unet.torch import dataset as D
    noisy, clean = _batch(3, 40, 48, 3, seed=2)
    (tmp_path / "c").mkdir(), (tmp_path / "n").mkdir()
    for i in range(3):
        Image.fromarray(noisy[i][:, :, ::-1]).save(tmp_path / "n" / f"{i}.png")
        Image.fromarray(clean[i][:, :, ::-1]).save(tmp_path / "c" / f"{i}.png")
    ds = D.DenoisingDataset(["0", "1", "2"], augments=D.DenoisingBatchTransform(32, train=False), clean_folder=str(tmp_path / "c"),
                            noisy_folder=str(tmp_path / "n"), im_size=32)
    (b,) = list(D.DeviceBatches(ds, batch_size=3))
    rx, ry = O.denoise_prepare(noisy, clean, 32, None)
    assert torch.equal(b["x"].cpu(), rx) and torch.equal(b["y"].cpu(), ry)
