"""Generate the golden fixtures under tests/golden/ by driving the *unmodified* reference code.

Run in the build container only (`python tests/golden/make_golden.py`): it imports
`/root/reference/vit_unet/torch/model.py`, which never travels to the GPU box.  What is committed
is data: inputs, expected outputs, losses, sampled gradients, checksums.

How the reference is driven:
  * `model.py:3,5` import `torchvision` and (via functions.py:2) `skimage`, both absent from this
    image.  Two inert in-memory modules satisfy the import statements; the only symbol the path
    touches is `torchvision.transforms.Resize`, which is the identity whenever the input is
    already im_size x im_size (spec decision D4) - the fixture inputs always are.
  * `HViT_UNet.__init__` raises at model.py:309, so objects are assembled with `__new__` +
    `nn.Module.__init__` and the ctor loops of model.py:310-370 re-run with the reference's own
    block classes.  `PatchEncoder.forward` and `HViT_UNet.forward` then run UNMODIFIED.
  * Weights come from the oracle's deterministic generator and are loaded with `load_state_dict`.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vit_unet_oracle as O  # noqa: E402

REF = "/root/reference"


def import_reference():
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")

        class Resize:  # identity for square inputs of the right size (D4)
            def __init__(self, size):
                self.size = size

            def __call__(self, x):
                assert x.shape[-1] == self.size and x.shape[-2] == self.size
                return x
        tr.Resize = Resize
        tv.transforms = tr
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tr
    if "skimage" not in sys.modules:
        sk = types.ModuleType("skimage")
        skm = types.ModuleType("skimage.metrics")
        skm.peak_signal_noise_ratio = None
        sk.metrics = skm
        sys.modules["skimage"] = sk
        sys.modules["skimage.metrics"] = skm
    sys.path.insert(0, REF)
    try:
        for k in [k for k in sys.modules if k == "vit_unet" or k.startswith("vit_unet.")]:
            del sys.modules[k]
        ref = importlib.import_module("vit_unet.torch.model")
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "vit_unet" or k.startswith("vit_unet.")]:
            del sys.modules[k]
    return ref


def build_reference(ref, cfg: O.Config):
    """Assemble a reference HViT_UNet without calling its broken ctor."""
    nn = torch.nn
    pe = ref.PatchEncoder.__new__(ref.PatchEncoder)
    nn.Module.__init__(pe)
    N0, D0, _, _ = cfg.level(0)
    pe.patch_size, pe.num_channels = cfg.patch_size, cfg.num_channels
    pe.positions = torch.arange(N0)
    pe.position_embedding = nn.Embedding(N0, D0)
    m = ref.HViT_UNet.__new__(ref.HViT_UNet)
    nn.Module.__init__(m)
    m.depth, m.depth_te, m.im_size = cfg.depth, cfg.depth_te, cfg.im_size
    m.num_channels, m.preprocessing, m.verbose = cfg.num_channels, cfg.preprocessing, False
    m.PE = pe
    TE = ref.ReAttentionTransformerEncoder

    def te(l):
        N, D, hid, _ = cfg.level(l)
        return TE(N, cfg.num_channels, D, hid, cfg.num_heads, cfg.attn_drop, cfg.proj_drop, cfg.linear_drop)
    m.Encoders = nn.ModuleList([te(l) for l in range(cfg.depth) for _ in range(cfg.depth_te)])
    m.BottleNeck = nn.ModuleList([te(cfg.depth) for _ in range(cfg.size_bottleneck)])
    m.Decoders = nn.ModuleList([te(cfg.depth - l) for l in range(cfg.depth) for _ in range(cfg.depth_te)])
    m.SkipConnections = nn.ModuleList([
        ref.SkipConnection(dim=cfg.level(cfg.depth - l - 1)[1], num_channels=cfg.num_channels,
                           num_heads=cfg.num_heads, attn_drop=cfg.attn_drop, proj_drop=cfg.proj_drop)
        for l in range(cfg.depth)])
    if cfg.preprocessing == "conv":
        m.conv2d = nn.Conv2d(cfg.num_channels, cfg.num_channels, 3, padding="same")
    return m


TINY = {
    # (i) tiny configs, full tensors
    "tiny_a": dict(depth=1, depth_te=1, size_bottleneck=1, preprocessing="conv", im_size=32,
                   patch_size=8, num_channels=3, hidden_dim=16, num_heads=2),
    "tiny_b": dict(depth=2, depth_te=1, size_bottleneck=1, preprocessing="conv", im_size=64,
                   patch_size=16, num_channels=1, hidden_dim=16, num_heads=2),
    # odd token count (49 tokens, as Base level 0) and head_dim 12 at the deepest level (as Lite L2)
    "tiny_c": dict(depth=1, depth_te=2, size_bottleneck=1, preprocessing="conv", im_size=56,
                   patch_size=8, num_channels=3, hidden_dim=8, num_heads=4),
}
GRAD_NAMES = ["PE.position_embedding.weight", "Encoders.0.ReAttn.qconv2d.weight",
              "Encoders.0.ReAttn.reatten_matrix.weight", "Encoders.0.ReAttn.var_norm.weight",
              "Encoders.0.ReAttn.proj.weight", "Encoders.0.LN1.weight", "Encoders.0.LN2.bias",
              "Encoders.0.FeedForward.net.0.weight", "BottleNeck.0.ReAttn.kconv2d.weight",
              "BottleNeck.0.FeedForward.net.3.bias", "SkipConnections.0.vconv2d.weight",
              "SkipConnections.0.reatten_matrix.bias", "SkipConnections.0.proj.bias",
              "conv2d.weight", "conv2d.bias"]


def run_reference(ref, cfg, w, x, y, training):
    m = build_reference(ref, cfg)
    missing = m.load_state_dict({k: v.clone() for k, v in w.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    m.train(training)
    out = m(x)
    loss = torch.nn.MSELoss()(out, y)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    bufs = {k: b.detach().clone() for k, b in m.named_buffers()}
    return out.detach(), loss.detach(), grads, bufs


FULL_TRAIN_GRADS = ["PE.position_embedding.weight", "Encoders.0.ReAttn.qconv2d.weight",
                    "Encoders.0.ReAttn.reatten_matrix.weight", "Encoders.0.ReAttn.var_norm.weight",
                    "Encoders.0.ReAttn.proj.weight", "Encoders.0.LN1.weight", "BottleNeck.0.ReAttn.kconv2d.weight",
                    "BottleNeck.0.ReAttn.reatten_matrix.weight", "BottleNeck.0.FeedForward.net.0.weight",
                    "Decoders.0.LN2.bias", "Decoders.0.ReAttn.var_norm.bias", "SkipConnections.0.vconv2d.weight",
                    "SkipConnections.1.proj.weight", "conv2d.weight", "conv2d.bias"]


def run_reference_dtype(ref, cfg, B, dtype):
    w = O.make_weights(cfg, seed=0, dtype=dtype)
    x, y = O.make_batch(cfg, B=B, seed=1234, dtype=dtype)
    m = build_reference(ref, cfg)
    if dtype == torch.float64:
        m = m.double()
    m.load_state_dict({k: v.clone() for k, v in w.items()}, strict=True)
    m.train(True)
    out = m(x)
    loss = torch.nn.MSELoss()(out, y)
    loss.backward()
    grads = {k: p.grad.detach().double() for k, p in m.named_parameters()}
    bufs = {k: b.detach().double() for k, b in m.named_buffers() if b.dtype.is_floating_point}
    return out.detach().double(), float(loss), grads, bufs


def full_train(ref):
    """(v) full-size configs in TRAIN mode (BatchNorm batch statistics, all dropout 0).

    The full-depth train-mode model is ill-conditioned in float32 (measured here: the reference's OWN float32 run
    differs from its float64 run by 2e-2 in the output and 0.89 in gradient cosine for Base - twelve BatchNorms over
    attention maps amplify rounding by ~2.6x each), so the fixture holds the reference driven in FLOAT64 (the truth:
    loss, 64 sampled outputs, per-parameter gradient L1, <= 64 sampled elements of named gradients, BatchNorm running
    statistics) AND how far the reference's float32 run is from it (`ref32.*`): tests hold an fp32 implementation to
    a multiple of the reference's own float32 deviation.  Weights / batch are regenerated from seeds by the tests."""
    blob, meta = {}, {}
    for name, B in (("base", 2), ("large", 2), ("lite", 1)):
        kw = dict(O.PRESETS[name], attn_drop=0.0, proj_drop=0.0)
        cfg = O.Config(**kw)
        out, loss, grads, bufs = run_reference_dtype(ref, cfg, B, torch.float64)
        out32, loss32, grads32, _ = run_reference_dtype(ref, cfg, B, torch.float32)
        flat = out.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        blob[f"{name}.out_idx"], blob[f"{name}.out_sample"] = idx.numpy(), flat[idx].numpy()
        blob[f"{name}.out_absmax"] = np.array(float(out.abs().max()))
        blob[f"{name}.loss"] = np.array(loss)
        blob[f"{name}.gradabs"] = np.array([float(grads[k].abs().sum()) for k, _ in O.param_shapes(cfg)])
        cos32 = {}
        for gname in FULL_TRAIN_GRADS:
            g = grads[gname].reshape(-1)
            gi = torch.linspace(0, g.numel() - 1, min(64, g.numel())).long()
            blob[f"{name}.grad_idx.{gname}"] = gi.numpy()
            blob[f"{name}.grad.{gname}"] = g[gi].numpy()
            blob[f"{name}.gradmax.{gname}"] = np.array(float(g.abs().max()))
            h = grads32[gname].reshape(-1)
            cos32[gname] = float(g @ h / (g.norm() * h.norm()))
        for k, b in bufs.items():
            if k.endswith(("Encoders.0.ReAttn.var_norm.running_mean", "Encoders.0.ReAttn.var_norm.running_var",
                           "BottleNeck.0.ReAttn.var_norm.running_mean", "BottleNeck.0.ReAttn.var_norm.running_var",
                           "SkipConnections.1.var_norm.running_var")):
                blob[f"{name}.buf.{k}"] = b.numpy()
        ga = torch.cat([grads[k].reshape(-1) for k, _ in O.param_shapes(cfg)])
        gb = torch.cat([grads32[k].reshape(-1) for k, _ in O.param_shapes(cfg)])
        meta[name] = {"B": B, "weights_seed": 0, "batch_seed": 1234, "loss": loss,
                      "ref32": {"loss_rel": abs(loss32 - loss) / abs(loss),
                                "out_err": float((out32 - out).abs().max() / out.abs().max()),
                                "grad_cos_all": float(ga @ gb / (ga.norm() * gb.norm())),
                                "grad_cos": cos32}}
        print("full_train", name, loss, meta[name]["ref32"]["loss_rel"], meta[name]["ref32"]["out_err"],
              meta[name]["ref32"]["grad_cos_all"])
    np.savez_compressed(os.path.join(HERE, "full_train.npz"), **blob)
    return meta


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "full_train":      # only (v); merged into the existing manifest
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest = json.load(f)
        manifest["full_train"] = full_train(ref)
        with open(os.path.join(HERE, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1)
        return
    manifest = {"torch": torch.__version__, "cases": {}}

    # ---- (i) tiny configs: eval and train (dropout 0 -> BN batch-stat path), full tensors ----
    for name, kw in TINY.items():
        cfg = O.Config(**kw, attn_drop=0.0, proj_drop=0.0, linear_drop=0.0)
        w = O.make_weights(cfg, seed=7)
        x, y = O.make_batch(cfg, B=3, seed=1234)
        blob = {"x": x.numpy(), "y": y.numpy()}
        for mode in ("eval", "train"):
            out, loss, grads, bufs = run_reference(ref, cfg, w, x, y, training=(mode == "train"))
            blob[f"{mode}.out"] = out.numpy()
            blob[f"{mode}.loss"] = loss.numpy()
            for g in GRAD_NAMES:
                if g in grads:
                    blob[f"{mode}.grad.{g}"] = grads[g].numpy()
            gsum = {k: float(v.double().abs().sum()) for k, v in grads.items()}
            blob[f"{mode}.gradabs"] = np.array([gsum[k] for k, _ in O.param_shapes(cfg)])
            if mode == "train":
                for k, b in bufs.items():
                    if "running" in k:
                        blob[f"train.buf.{k}"] = b.numpy()
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **blob)
        manifest["cases"][name] = {"config": kw, "B": 3, "weights_seed": 7, "batch_seed": 1234,
                                   "params": O.param_count(cfg)}
        # state_dict key order / shapes must equal the oracle's table
        m = build_reference(ref, cfg)
        assert [(k, tuple(v.shape)) for k, v in m.named_parameters()] == \
            [(k, tuple(s)) for k, s in O.param_shapes(cfg)], name
        print(name, "params", O.param_count(cfg))

    # ---- (ii) per-op vectors ----
    ops = {}
    g = torch.Generator().manual_seed(99)
    X = torch.rand(2, 3, 32, 32, generator=g)
    pt = torch.flatten(ref.patch(X, 8), -3, -1)
    ops["patch.in"], ops["patch.out_s8"] = X.numpy(), pt.numpy()
    ops["unpatch.out"] = ref.unpatch(ref.unflatten(pt, 3), 3).reshape(2, 3, 32, 32).numpy()
    ops["down.out"] = ref.downsampling(pt, 3).numpy()
    ops["up.out"] = ref.upsampling(pt, 3).numpy()
    # ReAttention / SkipConnection / FeedForward / TE block at N=49 (odd), D=192, h=4 -> d=48;
    # and at N=16, D=48, h=4 -> d=12
    for tag, (N, C, s, h, hid) in {"n49": (49, 3, 8, 4, 16), "d12": (16, 3, 4, 4, 8)}.items():
        D = C * s * s
        torch.manual_seed(5)
        blk = ref.ReAttentionTransformerEncoder(N, C, D, hid, h, 0.0, 0.0, 0.0)
        skp = ref.SkipConnection(dim=D, num_channels=C, num_heads=h)
        with torch.no_grad():
            for prm in list(blk.parameters()) + list(skp.parameters()):
                prm.add_(0.05 * torch.randn(prm.shape, generator=g))
        xin = torch.randn(2, N, D, generator=g)
        enc = torch.randn(2, N, D, generator=g)
        import copy
        sd_blk, sd_skp = copy.deepcopy(blk.state_dict()), copy.deepcopy(skp.state_dict())
        for k, v in sd_blk.items():
            ops[f"{tag}.blk.{k}"] = v.numpy().copy()
        for k, v in sd_skp.items():
            ops[f"{tag}.skp.{k}"] = v.numpy().copy()
        for mode in ("eval", "train"):
            blk.train(mode == "train"); skp.train(mode == "train")
            # every vector is taken from the INITIAL state (train-mode calls advance BN buffers)
            blk.load_state_dict(sd_blk)
            a, amap = blk.ReAttn(xin)
            ops[f"{tag}.{mode}.attn.out"], ops[f"{tag}.{mode}.attn.map"] = a.detach().numpy(), amap.detach().numpy()
            blk.load_state_dict(sd_blk)
            ops[f"{tag}.{mode}.block.out"] = blk(xin).detach().numpy()
            skp.load_state_dict(sd_skp)
            ops[f"{tag}.{mode}.skip.out"] = skp(enc, xin, xin).detach().numpy()
        ops[f"{tag}.ff.out"] = blk.FeedForward(xin).detach().numpy()
        ops[f"{tag}.x"], ops[f"{tag}.enc"] = xin.numpy(), enc.numpy()
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)

    # ---- (iii) full configs: checksums + sampled outputs (weights regenerated, never stored) ----
    full = {}
    cfgs = dict(O.PRESETS)
    cfgs["seg512"] = dict(O.PRESETS["base"], im_size=512, num_channels=1)
    for name, kw in cfgs.items():
        kw0 = dict(kw, attn_drop=0.0, proj_drop=0.0)
        cfg = O.Config(**kw0)
        w = O.make_weights(cfg, seed=0)
        B = 1 if name == "seg512" else 2
        x, y = O.make_batch(cfg, B=B, seed=1234)
        m = build_reference(ref, cfg)
        m.load_state_dict({k: v.clone() for k, v in w.items()})
        m.eval()
        with torch.no_grad():
            out = m(x)
        flat = out.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        full[name] = {"params": O.param_count(cfg), "B": B,
                      "sum": float(out.double().sum()), "mean": float(out.double().mean()),
                      "absmax": float(out.abs().max()),
                      "sample_idx": idx.tolist(), "sample": flat[idx].double().tolist()}
        print(name, full[name]["params"], full[name]["mean"])
    manifest["full"] = full
    manifest["full_train"] = full_train(ref)
    manifest["kat"] = {"readme_counts": {"lite": 3387568, "base": 36613036, "large": 63043866},
                       "packaged_counts": {"lite": 5193820, "base": 39623512, "large": 69064902,
                                           "seg512": 14919406}}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
