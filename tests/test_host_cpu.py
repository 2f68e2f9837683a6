"""CPU-side checks of the boundary: the C-ABI library loads and exports every declared symbol,
the C parameter table equals the reference state_dict layout (via the oracle's table), workspace
sizing, error conventions.  No kernel is launched (no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

import vit_unet_oracle as O
from vit_unet.torch import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "vit_unet_amd.h")).read()
    names = set(re.findall(r"\b(vu_[a-z0-9_]+)\s*\(", hdr))
    names -= {"vu_config", "vu_param_entry", "vu_attn_params", "vu_attn_grads"}
    assert len(names) >= 20
    L = C.CDLL(_lib.LIB_PATH)
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/vit_unet_amd.h but not exported"
    assert set(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ names


def _c_struct_fields(text, name):
    """[(ctype, field)] of `typedef struct <name> { ... } <name>;` in a C header (comments stripped, `int a, b;` expanded)."""
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.replace("*", " * ").split())
        if not decl:
            continue
        first, *rest = decl.split(",")
        toks = first.split()
        tname = " ".join(t for t in toks[:-1] if t not in ("const", "*"))       # "int", "float", "long long", "void"
        for piece in [first] + rest:
            ptoks = piece.split()
            out.append(("ptr" if "*" in ptoks else tname, ptoks[-1].split("[")[0]))
    return out


def test_header_structs_match_ctypes_and_integration_stub():
    """Field count, order and type of every struct the C ABI passes by pointer: header vs vit_unet/torch/_lib.py vs the
    ctypes stub a maintainer would copy from INTEGRATION.md (a short struct makes the C side read past its end)."""
    hdr = open(os.path.join(ROOT, "include", "vit_unet_amd.h")).read()
    cmap = {"int": C.c_int, "float": C.c_float, "long long": C.c_longlong}
    for sname, cls in (("vu_config", _lib.vu_config), ("vu_attn_params", _lib.vu_attn_params), ("vu_attn_grads", _lib.vu_attn_grads)):
        want = _c_struct_fields(hdr, sname)
        got = [(n, t) for n, t in cls._fields_]
        assert [n for _, n in want] == [n for n, _ in got], (sname, want, got)
        for (ct, n), (_, t) in zip(want, got):
            assert t is (C.c_void_p if ct == "ptr" else cmap[ct]), (sname, n, ct, t)
    L = _lib.lib()
    assert L.vu_config_size() == C.sizeof(_lib.vu_config) and L.vu_version() == _lib.ABI_VERSION
    # the INTEGRATION.md stub: same field names in the same order, and a constructor call with as many arguments
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class vu_config(C.Structure)"):doc.index("lib = C.CDLL")]
    assert re.findall(r'\("([a-z_]+)", C\.c_', stub) == [n for n, _ in _lib.vu_config._fields_]
    call = re.search(r"cfg = vu_config\(([^)]*)\)", doc).group(1)
    assert len(call.split(",")) == len(_lib.vu_config._fields_)
    assert f"vu_version() == {_lib.ABI_VERSION}" in doc


@pytest.mark.parametrize("name", ["lite", "base", "large"])
def test_param_table_matches_reference_layout(name):
    kw = O.PRESETS[name]
    ocfg = O.Config(**kw)
    cfg = _lib.make_config(dtype=torch.float32, **kw)
    table = _lib.param_table(cfg)
    assert [(n, s) for n, _, s, _ in table] == [(n, tuple(s)) for n, s in O.param_shapes(ocfg)]
    # offsets: increasing, 8-element aligned, non-overlapping
    end = 0
    for n, off, shape, _ in table:
        assert off % 8 == 0 and off >= end
        numel = 1
        for d in shape:
            numel *= d
        end = off + numel
    assert _lib.lib().vu_model_param_elems(C.byref(cfg)) >= end
    assert sum(torch.Size(s).numel() for _, _, s, _ in table) == O.param_count(ocfg)
    n_attn = sum(1 for n, *_ in table if n.endswith("var_norm.weight"))
    assert _lib.lib().vu_model_num_attn(C.byref(cfg)) == n_attn
    assert sorted(b for *_, b in table if b >= 0) == list(range(n_attn))


def test_constructor_asserts_match_reference():
    L = _lib.lib()
    bad = [dict(depth=2, patch_size=6), dict(depth=3, patch_size=16), dict(im_size=100, patch_size=16)]
    for b in bad:
        kw = dict(O.PRESETS["lite"]); kw.update(b)
        cfg = _lib.make_config(dtype=torch.float32, **kw)
        with pytest.raises(AssertionError):
            _lib.check(L.vu_model_validate(C.byref(cfg)))
    from vit_unet.torch import model as M
    with pytest.raises(AssertionError):
        M.HViT_UNet(2, 1, 1, "conv", 224, 6, 3, 16, 2, 0., 0., 0.)
    with pytest.raises(ValueError):
        M.get_vit_unet("huge")
    with pytest.raises(NotImplementedError):
        M.HViT_UNet(1, 1, 1, "fourier", 32, 8, 3, 16, 2, 0., 0., 0.)


def test_workspace_grows_with_batch():
    L = _lib.lib()
    cfg = _lib.make_config(dtype=torch.bfloat16, **O.PRESETS["base"])
    w8, w64 = L.vu_model_workspace_bytes(C.byref(cfg), 8), L.vu_model_workspace_bytes(C.byref(cfg), 64)
    assert 0 < w8 < w64 < 16 * 2 ** 30
    cfg32 = _lib.make_config(dtype=torch.float32, **O.PRESETS["base"])
    assert L.vu_model_workspace_bytes(C.byref(cfg32), 8) > w8


def test_eval_workspace_leaves_the_probability_caches_out_and_the_budget_caps_them():
    """Round 6: the probability caches of the recompute attention are a TRAINING buffer - an eval workspace is sized
    without them - and one workspace never spends more than the process's budget on them (modules beyond it recompute)."""
    L = _lib.lib()
    cfg = _lib.make_config(dtype=torch.bfloat16, **O.PRESETS["base"])
    tr, ev = L.vu_model_workspace_bytes_ex(C.byref(cfg), 64, 1), L.vu_model_workspace_bytes_ex(C.byref(cfg), 64, 0)
    pc = L.vu_model_pcache_bytes(C.byref(cfg), 64)
    assert tr == L.vu_model_workspace_bytes(C.byref(cfg), 64)
    per_module = 64 * 49 * 49 * 4096                  # B (N / 16)^2 tiles of 4 KB at N = 784
    assert pc == 4 * per_module and tr - ev >= pc and tr - ev < pc + 4096
    try:
        _lib.check(L.vu_set_flash_pcache_budget(2 * per_module + 1))
        assert L.vu_model_pcache_bytes(C.byref(cfg), 64) == 2 * per_module
        assert L.vu_model_workspace_bytes(C.byref(cfg), 64) < tr
        _lib.check(L.vu_set_flash_pcache_budget(0))
        assert L.vu_model_pcache_bytes(C.byref(cfg), 64) == 0
        assert L.vu_model_workspace_bytes(C.byref(cfg), 64) == L.vu_model_workspace_bytes_ex(C.byref(cfg), 64, 0)
    finally:
        _lib.check(L.vu_set_flash_pcache_budget(96 << 30))
    # fp32 storage has no recompute form: nothing to leave out
    cfg32 = _lib.make_config(dtype=torch.float32, **O.PRESETS["base"])
    assert L.vu_model_pcache_bytes(C.byref(cfg32), 64) == 0


def test_module_surface_and_state_dict_keys():
    from vit_unet.torch import model as M
    m = M.get_vit_unet("lite")
    ocfg = O.Config(**O.PRESETS["lite"])
    assert [k for k, _ in m.named_parameters()] == [k for k, _ in O.param_shapes(ocfg)]
    sd = m.state_dict()
    want = [k for k, _ in O.param_shapes(ocfg)] + [k for k, _ in O.buffer_shapes(ocfg)]
    assert sorted(sd.keys()) == sorted(want)
    assert sum(p.numel() for p in m.parameters()) == 5193820
    for attr in ("PE", "Encoders", "BottleNeck", "Decoders", "SkipConnections", "conv2d"):
        assert hasattr(m, attr)
    # README ctor surface
    r = M.ViT_UNet(depth=2, depth_te=2, size_bottleneck=2, preprocessing="conv", num_patches=49, patch_size=32,
                   num_channels=3, hidden_dim=128, num_heads=8, attn_drop=.2, proj_drop=.2, linear_drop=0,
                   dtype=torch.float32)
    assert r.im_size == 224 and sum(p.numel() for p in r.parameters()) == 39623512


def test_product_path_refuses_cpu_tensors():
    from vit_unet.torch import model as M
    m = M.HViT_UNet(1, 1, 1, "conv", 32, 8, 3, 16, 2, 0., 0., 0.)
    with pytest.raises(_lib.VuError):
        m(torch.rand(1, 3, 32, 32))


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "vit-unet_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "vit_unet_oracle" not in txt or f == "vu_common.h" and "import" not in txt, f


def test_fitter_surface_and_checkpoint_roundtrip(tmp_path):
    """ImageFitter counterpart (reference dataset.py:76-91 / run_denoising.py:84-100): unpack, checkpoint files with the
    module's state_dict key names, load restores the weights.  (The batch step itself needs the GPU: tests/test_gpu_model.py.)"""
    import torch
    import vit_unet.torch.model as M
    from vit_unet.torch.fitter import ImageFitter
    kw = dict(depth=1, depth_te=1, size_bottleneck=1, preprocessing="conv", im_size=32, patch_size=8, num_channels=3,
              hidden_dim=16, num_heads=2, attn_drop=0.0, proj_drop=0.0, linear_drop=0.0)
    m = M.HViT_UNet(**kw)
    f = ImageFitter(m, device="cpu", folder=str(tmp_path))
    x, y, w = f.unpack({"x": torch.zeros(2, 3, 32, 32, dtype=torch.float64), "y": torch.ones(2, 3, 32, 32), "w": torch.ones(2)})
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and w.shape == (2,)
    assert f.unpack({"x": x, "y": y})[2] is None
    path = str(tmp_path / "best-checkpoint.bin")
    f.epoch, f.best_metric = 3, 0.25
    f.save(path)
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(1.0)
    g = ImageFitter(m, device="cpu", folder=str(tmp_path)).load(path)
    assert g.epoch == 3 and g.best_metric == 0.25
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k
    assert not f._fused_ok()          # CPU device: never the fused HIP step


def test_dft_matrices_give_the_real_part_of_fft2():
    """variants.fft2_real is C_N X C_D - S_N X S_D with these matrices (two GEMMs on the device): the identity against
    torch.fft on the host, and the symmetry that makes the operator its own adjoint."""
    import numpy as np
    from vit_unet.torch.variants import dft_matrices
    rng = np.random.default_rng(0)
    for n, d in ((49, 192), (7, 5), (64, 64)):
        cn, sn = dft_matrices(n)
        cd, sd = dft_matrices(d)
        assert np.array_equal(cn, cn.T) and np.array_equal(sn, sn.T)
        x = rng.standard_normal((n, d))
        ref = torch.fft.fft2(torch.from_numpy(x)).real.numpy()
        got = cn @ x @ cd - sn @ x @ sd
        assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max()


def test_tf_token_pool_index_formula():
    """The oracle follows Resampling 'max' / 'avg' (tf/functions.py:101-124) reshape by reshape; the HIP kernel uses the closed
    form out[b, w + s N/8] = pool(x[b, 8 w + 2 s + {0, 1, 4, 5}]) (csrc/vu_tfops.hip): the two must be the same map."""
    import vit_unet_oracle as O
    g = torch.Generator().manual_seed(0)
    for N in (16, 64, 256):
        x = torch.randn(2, N, 12, generator=g)
        for mode, red in (("max", lambda t: t.max(0).values), ("avg", lambda t: t.mean(0))):
            got = O.tf_token_pool4(x, mode)
            for m in range(N // 4):
                s_, w = divmod(m, N // 8)
                src = 8 * w + 2 * s_
                ref = red(torch.stack([x[:, src], x[:, src + 1], x[:, src + 4], x[:, src + 5]]))
                assert torch.allclose(got[:, m], ref, atol=1e-6), (N, mode, m)


def test_no_mixed_opcode_mfma_accumulator_chain_in_the_recompute_sweeps():
    """tools/probe/mfma_chain_probe.hip (round 4): on gfx950, as hipcc 7.2 schedules them, a v_mfma_f32_16x16x16_bf16 whose
    accumulator input is the result of a v_mfma_f32_16x16x32_bf16 a few instructions earlier (or the other way round) reads a
    stale accumulator - wrong sums, no diagnostic (profiles/r04_mfma_chain_probe.txt).  vu_flash.hip is the one file that uses
    both opcodes; its generated code must not contain that pattern (tools/mfma_chain_scan.py compiles it with hipcc -S)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_chain_scan.py"), os.path.join(root, "vit-unet_amd", "csrc", "vu_flash.hip")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MFMA_CHAIN_SCAN CLEAN" in r.stdout, r.stdout + r.stderr


def test_library_links_no_vendor_blas():
    """The product library depends on the HIP runtime only: no hipBLASLt / rocBLAS / MIOpen in its NEEDED entries (round 3 linked
    hipBLASLt for the big GEMMs; round 4 runs them on csrc/vu_bgemm.hip)."""
    import subprocess
    so = os.path.join(os.path.dirname(_lib.__file__), "libvitunet_amd.so")
    out = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    needed = [l.split("[")[1].split("]")[0] for l in out.splitlines() if "(NEEDED)" in l]
    assert needed, out
    assert not [n for n in needed if any(t in n.lower() for t in ("blas", "miopen", "tensile", "rccl"))], needed
