"""Parity of what bench.py measures: FULL-SIZE train steps (Lite / Base / Large), fp32 and bf16 storage.

What is asserted, and why in this form.  The reference's full-depth train-mode model is ill-conditioned in float32:
driven UNMODIFIED in float32 and in float64 on the same weights it disagrees with itself (tests/golden/manifest.json,
`full_train.*.ref32`): Lite 7e-6 in the output (well conditioned), Base 2e-2 / gradient cosine 0.89, Large 1.2 /
cosine -0.53 (chaotic: twenty BatchNorms over attention maps).  So

  * the reference driven in FLOAT64 is the truth (full_train.npz); the CPU oracle reproduces it to 1e-7 in float64
    (tests/test_oracle_golden.py), which pins the algorithm at full size;
  * Lite is held to tight float32 tolerances on the loss, the output and ALL gradients (with and without dropout);
  * Base is held to a multiple of the reference's own float32 deviation, gradient by gradient against the float64
    oracle, and tightly on the well-conditioned last layers;
  * every transformer block and every skip module of Lite / Base / Large is checked at FULL dimensions, teacher-forced:
    it is fed the oracle's input for that block and compared - output, input gradient, every parameter gradient - with
    the oracle following the same bf16 rounding points (3e-2 forward / 5e-2 backward, scaled max error).  One block
    is well conditioned, so this is where a kernel error would show; the blocks run through the MODEL EXECUTOR
    (vu_model_forward / vu_model_backward of a one-block model of that level's shape), i.e. the benchmarked code path;
  * the bf16 and fp32 train steps are run side by side for 50 optimizer steps on one batch: the loss curves must agree.
"""
import json
import os

import numpy as np
import pytest
import torch

import vit_unet_oracle as O
import ctypes as C

from vit_unet.torch import _lib
from vit_unet.torch import model as M
from vit_unet.torch.engine import TrainStep

pytestmark = pytest.mark.gpu
DEV = "cuda"


def serr(got, ref):
    got = torch.as_tensor(np.asarray(got.detach().cpu() if torch.is_tensor(got) else got)).double()
    ref = torch.as_tensor(np.asarray(ref.detach().cpu() if torch.is_tensor(ref) else ref)).double()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def cosine(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()


def load_full_train(golden_dir):
    with open(os.path.join(golden_dir, "manifest.json")) as f:
        man = json.load(f)
    return man["full_train"], dict(np.load(os.path.join(golden_dir, "full_train.npz")))


def build(kw, weights, dtype=torch.float32):
    m = M.HViT_UNet(dtype=dtype, **kw)
    r = m.load_state_dict({k: v.clone().float() for k, v in weights.items()}, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    return m.to(DEV)


# ------------------------------------------------------------------------------------------------
# (a) fp32 HIP train step against the reference's float64 run (golden) + the float64 oracle for ALL gradients
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["lite", "base", "large"])
def test_full_config_train_fp32_vs_reference(golden_dir, name):
    meta, g = load_full_train(golden_dir)
    meta = meta[name]
    r32 = meta["ref32"]
    kw = dict(O.PRESETS[name], attn_drop=0.0, proj_drop=0.0)
    cfg = O.Config(**kw)
    w = O.make_weights(cfg, seed=meta["weights_seed"])
    x, y = O.make_batch(cfg, B=meta["B"], seed=meta["batch_seed"])
    m = build(kw, w).train()
    out = m(x.to(DEV))
    loss = torch.nn.MSELoss()(out, y.to(DEV))
    loss.backward()
    assert torch.isfinite(out).all()
    sd = dict(m.named_parameters())
    got_out = out.detach().reshape(-1)[torch.from_numpy(g[f"{name}.out_idx"]).to(DEV)].double().cpu().numpy()
    out_err = np.abs(got_out - g[f"{name}.out_sample"]).max() / float(g[f"{name}.out_absmax"])
    loss_rel = abs(loss.item() - meta["loss"]) / abs(meta["loss"])
    names = [k for k, _ in O.param_shapes(cfg)]

    def sampled_err(pname):
        got = sd[pname].grad.reshape(-1)[torch.from_numpy(g[f"{name}.grad_idx.{pname}"]).to(DEV)].double().cpu().numpy()
        return np.abs(got - g[f"{name}.grad.{pname}"]).max() / (float(g[f"{name}.gradmax.{pname}"]) + 1e-30)
    sampled = [k[len(f"{name}.grad."):] for k in g if k.startswith(f"{name}.grad.")]
    print(f"full train fp32 {name}: loss_rel {loss_rel:.3e} (ref32 {r32['loss_rel']:.3e}) out_err {out_err:.3e} (ref32 {r32['out_err']:.3e})")
    if name == "lite":                      # well conditioned: tight float32 statement on everything in the fixture
        assert loss_rel < 1e-4 and out_err < 5e-4, (loss_rel, out_err)
        for p in sampled:
            if not p.endswith("reatten_matrix.bias"):
                assert sampled_err(p) < 5e-3, p
        gabs = np.array([float(sd[k].grad.double().abs().sum()) for k in names])
        sel = np.array([not k.endswith("reatten_matrix.bias") for k in names])
        np.testing.assert_allclose(gabs[sel], g[f"{name}.gradabs"][sel], rtol=2e-2, atol=5e-5 * g[f"{name}.gradabs"].max())
        for k in g:
            if k.startswith(f"{name}.buf."):
                np.testing.assert_allclose(dict(m.named_buffers())[k[len(f"{name}.buf."):]].cpu().numpy(), g[k], rtol=2e-3)
        return
    if name == "large":                     # chaotic in the reference itself (ref32: output error 1.2, loss 18 % off, gradient
        assert r32["out_err"] > 0.1         # cosine -0.53): nothing but finiteness can be asserted on the whole model; the
        assert np.isfinite(loss.item())     # per-block statement for Large is the teacher-forced test below
        for k, p in sd.items():
            assert torch.isfinite(p.grad).all(), k
        return
    # base: a multiple of the reference's own float32 deviation
    assert loss_rel < 5e-2, (loss_rel, r32["loss_rel"])
    assert out_err < 8 * r32["out_err"], (out_err, r32["out_err"])
    # gradients against the REFERENCE's float64 run: the fixture holds sampled elements of 15 parameters' gradients (round 6: until then
    # this test re-ran the float64 oracle of the whole model for a cosine over all elements - a minute of CPU time, two on a slow box,
    # for a bound of 0.3; the oracle itself is pinned against the same fixture in the CPU suite)
    ga = np.concatenate([sd[p].grad.reshape(-1)[torch.from_numpy(g[f"{name}.grad_idx.{p}"]).to(DEV)].double().cpu().numpy() / (float(g[f"{name}.gradmax.{p}"]) + 1e-30)
                         for p in sampled if not p.endswith("reatten_matrix.bias")])
    gb = np.concatenate([g[f"{name}.grad.{p}"] / (float(g[f"{name}.gradmax.{p}"]) + 1e-30) for p in sampled if not p.endswith("reatten_matrix.bias")])
    call = float(np.dot(ga, gb) / (np.linalg.norm(ga) * np.linalg.norm(gb) + 1e-300))
    print(f"full train fp32 base: cosine of the sampled gradient elements (each tensor scaled by its max) vs the reference's float64 run: {call:.4f} (ref32 over all elements {r32['grad_cos_all']:.4f})")
    # (all three - this path, the reference's float32 run, the oracle's float32 run - are fp32 trajectories of a chaotic map: the
    # bound only catches a regression)
    assert call > 0.3, (call, r32["grad_cos_all"])
    for k in ("conv2d.weight", "conv2d.bias", "SkipConnections.1.proj.weight"):      # well-conditioned last layers (reference float32: 0.9998+)
        a_ = sd[k].grad.reshape(-1)[torch.from_numpy(g[f"{name}.grad_idx.{k}"]).to(DEV)].double().cpu().numpy()
        b_ = g[f"{name}.grad.{k}"]
        assert float(np.dot(a_, b_) / (np.linalg.norm(a_) * np.linalg.norm(b_) + 1e-300)) > 0.99, k


# ------------------------------------------------------------------------------------------------
# (b) full-size fp32 train step WITH dropout (hash replayed by the oracle), all gradients
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["lite", "base"])
def test_full_config_train_dropout_fp32_vs_oracle(name):
    B = 1                                            # (round 6: Base too - one image and the float32 oracle: its bounds below only catch a broken kernel)
    kw = dict(O.PRESETS[name])                       # attn_drop = proj_drop = 0.2 as the presets have it
    cfg = O.Config(**kw)
    dt = torch.float32
    w = O.make_weights(cfg, seed=0)
    x, y = O.make_batch(cfg, B=B, seed=1234)
    m = build(kw, w).train()
    m._step_seed = 2024
    out = m(x.to(DEV))
    loss = torch.nn.MSELoss()(out, y.to(DEV))
    loss.backward()
    wr = O.make_weights(cfg, seed=0, dtype=dt)
    names = [k for k, _ in O.param_shapes(cfg)]
    for k in names:
        wr[k].requires_grad_(True)
    ref = O.forward(wr, cfg, x.to(dt), training=True, seed=2024)
    lr = O.mse_loss(ref, y.to(dt))
    lr.backward()
    sd = dict(m.named_parameters())
    print(f"full train dropout {name}: out err {serr(out, ref):.3e} loss rel {abs(loss.item() - lr.item()) / abs(lr.item()):.3e}")
    if name == "lite":
        # with dropout Lite is no longer well conditioned either: the ORACLE's float32 and float64 runs differ by 1.0e-2
        # in the output (measured); the HIP fp32 path is held to the same order, gradients to their direction
        assert serr(out, ref) < 3e-2 and abs(loss.item() - lr.item()) < 1e-2 * abs(lr.item())
        ga = torch.cat([sd[k].grad.double().cpu().reshape(-1) for k in names if not k.endswith("reatten_matrix.bias")])
        gb = torch.cat([wr[k].grad.double().reshape(-1) for k in names if not k.endswith("reatten_matrix.bias")])
        print(f"full train dropout lite: gradient cosine {cosine(ga, gb):.5f}")
        assert cosine(ga, gb) > 0.99
        for k in ("conv2d.weight", "conv2d.bias", "SkipConnections.1.proj.weight"):
            assert serr(sd[k].grad, wr[k].grad) < 6e-2, k      # the output itself is 1e-2 off (fp32 vs fp64 oracle: same)
        return
    # base (float64 oracle): chaotic at full depth (section 2 of DESIGN.md): an ulp-level change of one kernel (another fma
    # contraction after a recompile) moves the float32 output by 0.1 - 0.25 of its range against float64, the reference's
    # own float32 run by 2e-2 without dropout; the bound only catches a broken kernel, the per-block statement is (c)
    assert serr(out, ref) < 0.5 and abs(loss.item() - lr.item()) < 1e-2 * abs(lr.item())
    ga = torch.cat([sd[k].grad.double().cpu().reshape(-1) for k in names if not k.endswith("reatten_matrix.bias")])
    gb = torch.cat([wr[k].grad.double().reshape(-1) for k in names if not k.endswith("reatten_matrix.bias")])
    # with dropout the whole-model gradient direction of Base is not reproducible in float32 at all (measured against the
    # float64 oracle: +0.4 in one build, -0.03 in the next after an unrelated recompile): printed, not asserted.  The float32
    # statement with dropout at full size is the teacher-forced one below (dtype float32), block by block.
    print(f"full train dropout base: gradient cosine vs the float32 oracle {cosine(ga, gb.double()):.4f}")
    for k in ("conv2d.weight", "conv2d.bias"):
        assert cosine(sd[k].grad, wr[k].grad) > 0.9, (k, cosine(sd[k].grad, wr[k].grad))


# ------------------------------------------------------------------------------------------------
# (c) teacher-forced blocks at full dimensions, bf16 storage, through the model executor
# ------------------------------------------------------------------------------------------------
BLOCK_KEYS = ["ReAttn.reatten_matrix.weight", "ReAttn.reatten_matrix.bias", "ReAttn.var_norm.weight", "ReAttn.var_norm.bias",
              "ReAttn.qconv2d.weight", "ReAttn.kconv2d.weight", "ReAttn.vconv2d.weight", "ReAttn.proj.weight", "ReAttn.proj.bias",
              "LN1.weight", "LN1.bias", "LN2.weight", "LN2.bias", "FeedForward.net.0.weight", "FeedForward.net.0.bias",
              "FeedForward.net.3.weight", "FeedForward.net.3.bias"]
BN_BUFS = ["ReAttn.var_norm.running_mean", "ReAttn.var_norm.running_var"]


# BASELINE config 5 as written: the Base constructor at 512x512x1, dropout 0.2 / 0.2, q, k, v rounded to OCP e4m3 (fp8
# attention operands) - levels (N, d) = (256, 128), (1024, 32), (4096, 8)
SEG512 = dict(O.PRESETS["base"], im_size=512, num_channels=1, attn_operands="e4m3")


def _preset(name):
    return SEG512 if name == "seg512" else O.PRESETS[name]


def _taps(cfg, B, seed):
    w = O.make_weights(cfg, seed=0)
    x, _ = O.make_batch(cfg, B=B, seed=1234)
    taps = {}
    tcfg = cfg
    if cfg.im_size > 224:
        # 512x512: the inputs of the blocks come from a forward WITHOUT dropout (the numpy replay of the pair-scheme mask over
        # four 8 x 4096 x 4096 maps alone takes minutes on the CPU); the blocks themselves are then run with dropout on
        tcfg = O.Config(**dict(cfg.__dict__, attn_drop=0.0, proj_drop=0.0))
    with torch.no_grad():
        O.forward({k: v.clone() for k, v in w.items()}, tcfg, x, training=True, seed=seed, taps=taps)
    return w, taps


def _one_block_model(cfg, lvl, dtype):
    N, D, hid, s = cfg.level(lvl)
    return M.HViT_UNet(depth=0, depth_te=1, size_bottleneck=1, preprocessing="none", im_size=cfg.im_size, patch_size=s,
                       num_channels=cfg.num_channels, hidden_dim=hid, num_heads=cfg.num_heads, attn_drop=cfg.attn_drop,
                       proj_drop=cfg.proj_drop, linear_drop=0.0, dtype=dtype, attn_operands=cfg.attn_operands).to(DEV).train()


def test_level0_block_at_a_token_count_that_takes_the_big_tile_kernel(attn_form, monkeypatch):
    """The level-0 block of Base (49 tokens x 3072 features) at 16 images = 784 token rows: its 3072 x 3072 projection runs
    on csrc/vu_bgemm.hip with bias + projection dropout + block residual fused in the epilogue, the data gradient and the
    fp32-accumulating weight gradient (ragged K = 784 = 12 k-steps + 16: 48 tail k-slots zeroed) with the deterministic
    bias-gradient sums behind it (at the 2 images of the teacher-forced test above the products stay on vu_gemm).  Same block, same bounds: output 3e-2,
    dx and every parameter gradient 5e-2 of their range against the oracle's bf16-storage restatement."""
    monkeypatch.setattr(O, "FLASH_FILL_RULE", False)
    dt, B, seed, lvl = torch.bfloat16, 16, 4321, 0
    cfg = O.Config(**O.PRESETS["base"])
    w = O.make_weights(cfg, seed=0)
    N, D, hid, s = cfg.level(lvl)
    pre = [k for k in w if k.endswith("ReAttn.proj.weight") and tuple(w[k].shape) == (D, D) and "Skip" not in k][0][:-len("ReAttn.proj.weight")]
    assert D == 3072 and B * N >= 512
    m = _one_block_model(cfg, lvl, dt)
    sdict = {"PE.position_embedding.weight": torch.zeros(N, D)}
    for k in BLOCK_KEYS + BN_BUFS:
        sdict["BottleNeck.0." + k] = w[pre + k].clone()
    sdict["BottleNeck.0.ReAttn.var_norm.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    m.load_state_dict(sdict, strict=True)
    m.zero_grad(set_to_none=False)
    m._shadow_clean = False
    m._step_seed = seed
    gen = torch.Generator().manual_seed(9)
    xin = torch.randn(B, N, D, generator=gen).to(dt).float()
    G = torch.randn(B, N, D, generator=gen).to(dt).float()
    C_ = cfg.num_channels
    L = _lib.lib()
    L.vu_prof_enable(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    X = O.unpatchify(xin, C_).to(DEV).requires_grad_(True)
    out = m(X)
    out.backward(O.unpatchify(G, C_).to(DEV))
    torch.cuda.synchronize()
    rep = json.loads(L.vu_prof_report().decode())
    assert ("bg_colsum_kernel" in rep and any(k.startswith("bgemm_kernel<NN,bf16,") for k in rep)
            and any(k.startswith("bgemm_kernel<NT,bf16,") for k in rep) and "bgemm_kernel<TT,f32 acc,224x192>" in rep), rep.keys()
    assert not any("hipblaslt" in k or k.startswith("Cijk") for k in rep), rep.keys()
    wr = {pre + k: w[pre + k].clone().requires_grad_(True) for k in BLOCK_KEYS}
    for k in BN_BUFS:
        wr[pre + k] = w[pre + k].clone()
    xr = xin.clone().requires_grad_(True)
    ref = O.te_block(xr, wr, pre, cfg, training=True, seed=seed, stream=0, storage=torch.bfloat16)
    (ref * G).sum().backward()
    assert serr(O.patchify(out.detach().cpu(), s), ref) < 3e-2
    assert serr(O.patchify(X.grad.cpu(), s), xr.grad) < 5e-2
    sd = dict(m.named_parameters())
    for k in BLOCK_KEYS:
        if k.endswith("reatten_matrix.bias"):
            continue                                        # analytically zero in train mode
        assert serr(sd["BottleNeck.0." + k].grad, wr[pre + k].grad) < 5e-2, k


@pytest.mark.parametrize("name,dt,ks", [("base", torch.bfloat16, 1), ("large", torch.bfloat16, 1), ("lite", torch.bfloat16, 1),
                                        ("base", torch.float32, 1), ("seg512", torch.bfloat16, 1), ("base", torch.bfloat16, 2)])
def test_teacher_forced_blocks_bf16_full_size(name, dt, ks, attn_form, monkeypatch):
    teacher_forced_blocks(name, dt, ks, attn_form, monkeypatch)


def teacher_forced_blocks(name, dt, ks, attn_form, monkeypatch, one_per_level=False):
    """Body of the teacher-forced block test.  `one_per_level`: only the FIRST block met at each level (the hot-path file that
    collects first runs one block per level of every preset in about a minute; the full sweep over all blocks is the test above)."""
    # the benchmarked batch runs every covered level in the recompute ("flash") form; at this test's batch the fill rule
    # would pick the materialising kernels for most levels, so force the form the bench line is made of (ks = 1: the unsplit
    # sweeps every batch size runs by default; ks = 2: the opt-in split form, kept correct)
    attn_form(flash=1, key_split=ks)
    monkeypatch.setattr(O, "FLASH_FILL_RULE", False)
    B = 1                                            # (round 6: one image for every preset - the oracle on the host is the cost of this test)
    cfg = O.Config(**_preset(name))                  # dropout 0.2 / 0.2 as benchmarked
    seed = 777
    w, taps = _taps(cfg, B, seed)
    C_ = cfg.num_channels
    models = {}
    gen = torch.Generator().manual_seed(5)
    worst = {"fwd": 0.0, "bwd": 0.0}
    nrun = 0
    for pre, xin, lvl, _stream in taps["blocks"]:
        N, D, hid, s = cfg.level(lvl)
        if one_per_level and lvl in models:
            continue
        nrun += 1
        if lvl not in models:
            models[lvl] = _one_block_model(cfg, lvl, dt)
        m = models[lvl]
        sdict = {"PE.position_embedding.weight": torch.zeros(N, D)}
        for k in BLOCK_KEYS + BN_BUFS:
            sdict["BottleNeck.0." + k] = w[pre + k].clone()
        sdict["BottleNeck.0.ReAttn.var_norm.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        r = m.load_state_dict(sdict, strict=True)
        assert not r.missing_keys and not r.unexpected_keys
        m.zero_grad(set_to_none=False)
        m._shadow_clean = False
        m._step_seed = seed
        G = torch.randn(B, N, D, generator=gen).to(torch.bfloat16).float()
        st_ = torch.bfloat16 if dt == torch.bfloat16 else None
        tol_f, tol_b = (3e-2, 5e-2) if dt == torch.bfloat16 else (2e-3, 1e-2)
        X = O.unpatchify(xin, C_).to(DEV).requires_grad_(True)
        out = m(X)
        out.backward(O.unpatchify(G, C_).to(DEV))
        torch.cuda.synchronize()
        # oracle: same input (as stored: bf16), same rounding points, the executor's dropout stream 0
        wr = {pre + k: w[pre + k].clone().requires_grad_(True) for k in BLOCK_KEYS}
        for k in BN_BUFS:
            wr[pre + k] = w[pre + k].clone()
        xr = (xin.to(torch.bfloat16).float() if dt == torch.bfloat16 else xin.clone().float()).requires_grad_(True)
        ref = O.te_block(xr, wr, pre, cfg, training=True, seed=seed, stream=0, storage=st_)
        (ref * G).sum().backward()
        ef = serr(O.patchify(out.detach().cpu(), s), ref)
        assert ef < tol_f, (pre, "out", ef)
        eb = serr(O.patchify(X.grad.cpu(), s), xr.grad)
        assert eb < tol_b, (pre, "dx", eb)
        sd = dict(m.named_parameters())
        if os.environ.get("VU_TF_DEBUG"):
            print(pre, "out", f"{ef:.3e}", "dx", f"{eb:.3e}", " ".join(
                f"{k.replace('ReAttn.', '').replace('FeedForward.', 'FF.')}={serr(sd['BottleNeck.0.' + k].grad, wr[pre + k].grad):.2e}"
                for k in BLOCK_KEYS if not k.endswith("reatten_matrix.bias")), flush=True)
            gw, gbb = wr[pre + "ReAttn.var_norm.weight"].grad, wr[pre + "ReAttn.var_norm.bias"].grad
            print("   dgamma", gw.tolist(), "\n   dbeta ", gbb.tolist(), "\n   beta", wr[pre + "ReAttn.var_norm.bias"].tolist(),
                  "\n   hip dgamma", sd["BottleNeck.0.ReAttn.var_norm.weight"].grad.tolist(), flush=True)
            continue
        # (Round 3: the first decoder block of a level reads the un-normalised output of a SkipConnection - at 512 x 512 its
        # input has std ~80 and 92 % of the softmax rows are one-hot.  The recompute backward used to lose the q / k convolution
        # weight gradients there (errors of 100 - 2000 %): three fixes, DESIGN.md section 2 "saturated rows"; this test now holds
        # that block to the same bound as every other one.)
        for k in BLOCK_KEYS:
            if k.endswith("reatten_matrix.bias"):
                continue                                    # analytically zero in train mode (rounding noise only)
            e = serr(sd["BottleNeck.0." + k].grad, wr[pre + k].grad)
            if k.endswith("var_norm.weight") and e >= tol_b:
                # d gamma = (sum dA^ A^ - beta sum dA^) / gamma is formed from dO, O and v (no pass over the maps), with O as
                # stored (bf16: 2^-8).  Where |beta d beta| is orders above |d gamma| (seen at 512 x 512: 240 against 4) the
                # difference carries the rounding of the large terms: the bound is conditioning-aware there
                gam, bet = wr[pre + "ReAttn.var_norm.weight"].detach(), wr[pre + "ReAttn.var_norm.bias"].detach()
                big = (bet * wr[pre + "ReAttn.var_norm.bias"].grad / gam).abs().max().item()
                gmax = wr[pre + k].grad.abs().max().item()
                assert e * gmax < tol_b * gmax + 4e-3 * big, (pre, k, e, gmax, big)
                continue
            assert e < tol_b, (pre, k, e)
            eb = max(eb, e)
        # reatten_matrix.bias: |g| stays at the rounding-noise level of the mixing-matrix gradient
        gb = sd["BottleNeck.0.ReAttn.reatten_matrix.bias"].grad.abs().max().item()
        gw = sd["BottleNeck.0.ReAttn.reatten_matrix.weight"].grad.abs().max().item()
        assert gb < 2.0 * gw + 1e-6, (pre, gb, gw)
        bufs = dict(m.named_buffers())
        assert serr(bufs["BottleNeck.0.ReAttn.var_norm.running_var"], wr[pre + "ReAttn.var_norm.running_var"]) < 2e-2, pre
        worst["fwd"], worst["bwd"] = max(worst["fwd"], ef), max(worst["bwd"], eb)
    print(f"teacher-forced {name} {dt}: {nrun} of {len(taps['blocks'])} blocks, worst scaled error fwd {worst['fwd']:.3e} bwd {worst['bwd']:.3e}")


ATTN_KEYS = ["reatten_matrix.weight", "reatten_matrix.bias", "var_norm.weight", "var_norm.bias", "qconv2d.weight",
             "kconv2d.weight", "vconv2d.weight", "proj.weight", "proj.bias"]


@pytest.mark.parametrize("name", ["base", "lite", "seg512"])
def test_teacher_forced_skips_bf16_full_size(name, attn_form, monkeypatch):
    """The two SkipConnection modules at full dimensions (cross re-attention, q from the encoder), stand-alone module
    on bf16 tensors against the oracle with the same rounding points."""
    attn_form(flash=1)
    monkeypatch.setattr(O, "FLASH_FILL_RULE", False)
    B = 1
    cfg = O.Config(**_preset(name))
    seed = 778
    w, taps = _taps(cfg, B, seed)
    gen = torch.Generator().manual_seed(6)
    for pre, enc, dec, lvl, _stream in taps["skips"]:
        N, D, _, _ = cfg.level(lvl)
        skp = M.SkipConnection(dim=D, num_channels=cfg.num_channels, num_heads=cfg.num_heads, attn_drop=cfg.attn_drop,
                               proj_drop=cfg.proj_drop)
        sd = {k: w[pre + k].clone() for k in ATTN_KEYS}
        sd.update({"var_norm.running_mean": w[pre + "var_norm.running_mean"].clone(),
                   "var_norm.running_var": w[pre + "var_norm.running_var"].clone(),
                   "var_norm.num_batches_tracked": torch.zeros((), dtype=torch.int64)})
        skp.load_state_dict(sd)
        skp.attn_operands = cfg.attn_operands
        skp.to(DEV).train()
        G = torch.randn(B, N, D, generator=gen).to(torch.bfloat16)
        e16 = enc.to(torch.bfloat16).to(DEV).requires_grad_(True)
        d16 = dec.to(torch.bfloat16).to(DEV).requires_grad_(True)
        out = skp(e16, d16, d16, seed=seed, stream_id=3)
        out.backward(G.to(DEV))
        wr = {pre + k: w[pre + k].clone().requires_grad_(True) for k in ATTN_KEYS}
        wr[pre + "var_norm.running_mean"] = w[pre + "var_norm.running_mean"].clone()
        wr[pre + "var_norm.running_var"] = w[pre + "var_norm.running_var"].clone()
        er = enc.to(torch.bfloat16).float().requires_grad_(True)
        dr = dec.to(torch.bfloat16).float().requires_grad_(True)
        ref = O.skip_block(er, dr, wr, pre, cfg, training=True, seed=seed, stream=3, storage=torch.bfloat16)
        (ref * G.float()).sum().backward()
        assert serr(out.float(), ref) < 3e-2, pre
        assert serr(e16.grad.float(), er.grad) < 5e-2 and serr(d16.grad.float(), dr.grad) < 5e-2, pre
        ps = dict(skp.named_parameters())
        for k in ATTN_KEYS:
            if not k.endswith("reatten_matrix.bias"):
                assert serr(ps[k].grad, wr[pre + k].grad) < 5e-2, (pre, k)


# ------------------------------------------------------------------------------------------------
# (d) bf16 against fp32 over an optimisation trajectory (Base, 8 images, 50 fused steps on one batch)
# ------------------------------------------------------------------------------------------------
def test_bf16_vs_fp32_loss_trajectory_base():
    torch.manual_seed(0)
    m32 = M.get_vit_unet("base", dtype=torch.float32)
    m16 = M.get_vit_unet("base", dtype=torch.bfloat16)
    m16.load_state_dict(m32.state_dict())
    m32, m16 = m32.to(DEV).train(), m16.to(DEV).train()
    cfg = O.Config(**O.PRESETS["base"])
    x, y = O.make_batch(cfg, B=8, seed=99)
    x, y = x.to(DEV), y.to(DEV)
    t32, t16 = TrainStep(m32, lr=1e-3, seed=11), TrainStep(m16, lr=1e-3, seed=11)
    l32, l16 = [], []
    for _ in range(50):
        l32.append(t32.step(x, y).item())
        l16.append(t16.step(x, y).item())
    l32, l16 = np.array(l32), np.array(l16)
    rel = np.abs(l16 - l32) / l32
    print("trajectory fp32", np.round(l32, 3).tolist(), "\nbf16", np.round(l16, 3).tolist(), "\nmax rel diff", rel.max())
    assert np.isfinite(l16).all() and np.isfinite(l32).all()
    assert rel[0] < 0.1, rel[0]                        # the first step sees identical weights (full depth: chaotic forward)
    assert l32[-1] < 0.35 * l32[0] and l16[-1] < 0.35 * l16[0]   # both train
    w32, w16 = l32.reshape(5, 10).mean(axis=1), l16.reshape(5, 10).mean(axis=1)
    # and along the same curve, window by window: the fast drop around steps 10-17 starts a few steps apart in two fp32 /
    # bf16 runs (and between two runs of one precision: float atomics), which moves that window's mean by up to 35 %
    # (round 6: the two transition windows get a wider band.  The test is not deterministic - the fp32 path still has float atomics -
    # and with the drop starting at step 10 in one run and at step 14 in the other the third window's means were 0.52 and 0.83:
    # log ratio 0.47 - 0.55 from run to run of ONE build, against a bound of 0.5.  The windows before and after the drop keep 0.5.)
    lr_ = np.abs(np.log(w16 / w32))
    assert lr_[[0, 3, 4]].max() < 0.5 and lr_[[1, 2]].max() < 0.9, (w32, w16)
    assert abs(np.log(w16[-1] / w32[-1])) < 0.25, (w32, w16)


# ------------------------------------------------------------------------------------------------
# (e) surface that round 1 left untested: FeedForward alone, preprocessing='none', the ViT_UNet(...) alias
# ------------------------------------------------------------------------------------------------
def test_feedforward_module_vs_golden(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "ops.npz")))
    for tag, (N, C_, s, h, hid) in {"n49": (49, 3, 8, 4, 16), "d12": (16, 3, 4, 4, 8)}.items():
        D = C_ * s * s
        ff = M.FeedForward(D, hid, 0.0)
        ff.load_state_dict({k[len(f"{tag}.blk.FeedForward."):]: torch.from_numpy(v.copy()) for k, v in g.items()
                            if k.startswith(f"{tag}.blk.FeedForward.")})
        ff.to(DEV)
        x = torch.from_numpy(g[f"{tag}.x"]).to(DEV).requires_grad_(True)
        out = ff(x)
        assert serr(out, g[f"{tag}.ff.out"]) < 2e-5
        # backward against torch autograd on the same weights (CPU)
        ffc = torch.nn.Sequential(torch.nn.Linear(D, hid), torch.nn.GELU(), torch.nn.Dropout(0.0), torch.nn.Linear(hid, D),
                                  torch.nn.Dropout(0.0))
        ffc.load_state_dict({k: v.detach().cpu() for k, v in ff.net.state_dict().items()})
        xc = x.detach().cpu().requires_grad_(True)
        dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
        ffc(xc).backward(dy)
        out.backward(dy.to(DEV))
        assert serr(x.grad, xc.grad) < 2e-4
        for (k, p), (_, pc) in zip(ff.net.named_parameters(), ffc.named_parameters()):
            assert serr(p.grad, pc.grad) < 2e-4, k


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_feedforward_linear_drop_vs_oracle(dt):
    """linear_drop > 0 (model.py:102-108; 0 in every preset): both dropout sites replayed by the oracle."""
    B, N, D, hid = 2, 49, 192, 16
    gen = torch.Generator().manual_seed(8)
    ff = M.FeedForward(D, hid, 0.3).to(DEV).train()
    x = torch.randn(B, N, D, generator=gen)
    dy = torch.randn(B, N, D, generator=gen).to(dt)
    xd = x.to(dt).to(DEV).requires_grad_(True)
    out = ff(xd, seed=5, stream_id=2)
    out.backward(dy.to(DEV))
    p = {"net.0.weight": ff.net[0].weight.detach().cpu().to(dt).float().requires_grad_(True),
         "net.0.bias": ff.net[0].bias.detach().cpu().clone().requires_grad_(True),
         "net.3.weight": ff.net[3].weight.detach().cpu().to(dt).float().requires_grad_(True),
         "net.3.bias": ff.net[3].bias.detach().cpu().clone().requires_grad_(True)}
    xr = x.to(dt).float().requires_grad_(True)
    ref = O.feed_forward(xr, p, "", training=True, linear_drop=0.3, seed=5, stream=2, storage=dt)
    ref.backward(dy.float())
    ft, bt = (2e-5, 2e-4) if dt == torch.float32 else (3e-2, 5e-2)
    assert serr(out.float(), ref) < ft
    assert (out == 0).float().mean().item() > 0.2                      # the output dropout really dropped
    assert serr(xd.grad.float(), xr.grad) < bt
    for (k, q) in ff.net.named_parameters():
        assert serr(q.grad, p["net." + k].grad) < bt, k


@pytest.mark.parametrize("B,N,D,hid", [(8, 196, 192, 32), (3, 347, 192, 32), (64, 196, 192, 32), (4, 3136, 48, 16), (3, 347, 48, 16)])
def test_feedforward_fused_level2_vs_oracle(B, N, D, hid):
    """The level-2 shape (D = 192, hidden = 32, bf16, no linear dropout) takes the ONE-kernel-per-direction route of
    csrc/vu_ff2.hip (the launch profiler names it); forward and every gradient against the oracle's bf16-storage replay, on row
    counts that are and are not a multiple of the 16-token tile (3 * 347 = 1041).  (The residual form is the block's: the
    teacher-forced block tests and the whole-model parity tests run it.)"""
    import ctypes as C
    from vit_unet.torch._lib import lib
    dt = torch.bfloat16                                # (48, 16: Lite's 3136-token level, the generic form of the same kernel)
    gen = torch.Generator().manual_seed(21)
    ff = M.FeedForward(D, hid, 0.0).to(DEV).train()
    x = torch.randn(B, N, D, generator=gen)
    dy = torch.randn(B, N, D, generator=gen).to(dt)
    xd = x.to(dt).to(DEV).requires_grad_(True)
    L = lib()
    torch.cuda.synchronize()
    L.vu_prof_enable(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    out = ff(xd, seed=5, stream_id=2)
    out.backward(dy.to(DEV))
    torch.cuda.synchronize()
    rep = json.loads(L.vu_prof_report().decode())
    assert "vu_ff2_fwd_kernel" in rep and "vu_ff2_bwd_kernel" in rep, rep.keys()
    p = {"net.0.weight": ff.net[0].weight.detach().cpu().to(dt).float().requires_grad_(True),
         "net.0.bias": ff.net[0].bias.detach().cpu().clone().requires_grad_(True),
         "net.3.weight": ff.net[3].weight.detach().cpu().to(dt).float().requires_grad_(True),
         "net.3.bias": ff.net[3].bias.detach().cpu().clone().requires_grad_(True)}
    xr = x.to(dt).float().requires_grad_(True)
    ref = O.feed_forward(xr, p, "", training=True, linear_drop=0.0, seed=5, stream=2, storage=dt)
    ref.backward(dy.float())
    assert serr(out.float(), ref) < 1e-2                               # one bf16 rounding of the output
    assert serr(xd.grad.float(), xr.grad) < 2e-2
    for (k, q) in ff.net.named_parameters():
        assert serr(q.grad, p["net." + k].grad) < 2e-2, k


def test_model_linear_drop_vs_oracle(golden_dir):
    """A whole (tiny) model with all three dropouts on: element-for-element against the oracle's replay."""
    with open(os.path.join(golden_dir, "manifest.json")) as f:
        man = json.load(f)
    g = dict(np.load(os.path.join(golden_dir, "tiny_a.npz")))
    kw = dict(man["cases"]["tiny_a"]["config"], attn_drop=0.2, proj_drop=0.1, linear_drop=0.25)
    cfg = O.Config(**kw)
    w = O.make_weights(cfg, seed=7)
    m = build(kw, w).train()
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    m._step_seed = 99
    out = m(x)
    torch.nn.MSELoss()(out, y).backward()
    wr = {k: v.clone() for k, v in w.items()}
    for k, _ in O.param_shapes(cfg):
        wr[k].requires_grad_(True)
    ref = O.forward(wr, cfg, x.cpu(), training=True, seed=99)
    O.mse_loss(ref, y.cpu()).backward()
    assert serr(out, ref) < 2e-4
    sd = dict(m.named_parameters())
    for k, _ in O.param_shapes(cfg):
        if not k.endswith("reatten_matrix.bias"):
            assert serr(sd[k].grad, wr[k].grad) < 5e-3, k


def test_preprocessing_none_and_readme_alias(golden_dir):
    """preprocessing='none' (model.py:425: no output conv) and the README constructor `ViT_UNet(..., num_patches=...)`
    (README.md:18-31) run on the GPU against the oracle."""
    kw = dict(depth=1, depth_te=1, size_bottleneck=1, preprocessing="none", im_size=32, patch_size=8, num_channels=3,
              hidden_dim=16, num_heads=2, attn_drop=0.0, proj_drop=0.0, linear_drop=0.0)
    cfg = O.Config(**kw)
    w = O.make_weights(cfg, seed=3)
    assert "conv2d.weight" not in w
    x, y = O.make_batch(cfg, B=2, seed=4)
    for training in (False, True):
        m = build(kw, w).train(training)
        xd = x.to(DEV).requires_grad_(True)
        out = m(xd)
        torch.nn.MSELoss()(out, y.to(DEV)).backward()
        wr = {k: v.clone() for k, v in w.items()}
        for k, _ in O.param_shapes(cfg):
            wr[k].requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        ref = O.forward(wr, cfg, xr, training=training)
        O.mse_loss(ref, y).backward()
        assert serr(out, ref) < 2e-4
        assert serr(xd.grad, xr.grad) < 5e-3
        sd = dict(m.named_parameters())
        for k, _ in O.param_shapes(cfg):
            if not (training and k.endswith("reatten_matrix.bias")):
                assert serr(sd[k].grad, wr[k].grad) < 5e-3, k
    # README alias: num_patches instead of im_size; same arithmetic as HViT_UNet
    kc = dict(kw, preprocessing="conv")
    cfgc = O.Config(**kc)
    wc = O.make_weights(cfgc, seed=3)
    alias = M.ViT_UNet(depth=1, depth_te=1, size_bottleneck=1, preprocessing="conv", num_patches=16, patch_size=8,
                       num_channels=3, hidden_dim=16, num_heads=2, attn_drop=0.0, proj_drop=0.0, linear_drop=0.0,
                       dtype=torch.float32)
    assert isinstance(alias, M.HViT_UNet) and alias.im_size == 32
    alias.load_state_dict({k: v.clone() for k, v in wc.items()})
    alias.to(DEV).eval()
    with torch.no_grad():
        out = alias(x.to(DEV))
        ref = O.forward(wc, cfgc, x, training=False)
    assert serr(out, ref) < 2e-4


# ------------------------------------------------------------------------------------------------
# (f) the north-star statement: what the bf16 path does to the metrics the reference reports (functions.py:7-19 PSNR,
#     README.md:91-101 Dice), full-size models in eval mode against the fp32 CPU oracle.  bench.py prints the same
#     record in its JSON line (`parity`).
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,dt", [("lite", torch.bfloat16), ("base", torch.bfloat16), ("base", torch.float32),
                                     ("seg512", torch.bfloat16)])
def test_metric_deltas_full_size_eval(name, dt):
    import bench
    r = bench.eval_parity(name, dt, B=1 if name == "seg512" else 2, operands="e4m3" if name == "seg512" else "storage")
    print("eval parity", r)
    if dt == torch.float32:
        assert r["max_rel"] < 5e-4 and r["dpsnr"] < 1e-3, r
        return
    # bf16 storage (8 significant bits per stored activation, 12 attention modules deep) against the pure fp32 oracle:
    # stated bounds (measured: Base max_rel 9.3e-3, dPSNR 3.9e-3 dB)
    s_ = r["vs_same_rounding_points"]
    if name == "seg512":
        # fp8 (e4m3: 3 mantissa bits) attention operands move the random-init model's output by O(1) of its range against the
        # fp32 arithmetic - that is the format, reported in `r`; the kernels are held to the oracle that rounds where they do
        assert s_["max_rel"] < 0.1 and s_["rms_rel"] < 2e-2 and abs(s_["ddice"]) < 2e-3, r
        assert abs(r["ddice"]) < 5e-3, r
    else:
        assert r["max_rel"] < 0.1 and r["rms_rel"] < 2e-2 and r["dpsnr"] < 0.1, r            # dB
        assert s_["max_rel"] < 0.1, r
