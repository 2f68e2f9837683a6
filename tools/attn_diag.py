#!/usr/bin/env python3
"""Diagnostic: the recompute attention's INTERMEDIATE gradients (dq, dk, dv as the sweeps leave them in the workspace)
against the oracle's, for one attention module of the 512 x 512 x 1 configuration fed the oracle's own activations.
    python tools/attn_diag.py [--block Decoders.2.] [--operands e4m3] [--drop 0.2]
Used in round 3 to find why the q / k convolution weight gradients of the first level-1 decoder block were 100 - 400 %
off while every other gradient was within 1 %."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import vit_unet_oracle as O  # noqa: E402
from vit_unet.torch import _lib  # noqa: E402
from vit_unet.torch._lib import check, lib, ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--block", default="Decoders.2.")
ap.add_argument("--operands", default="e4m3")
ap.add_argument("--drop", type=float, default=0.2)
ap.add_argument("--flash", type=int, default=1)
a = ap.parse_args()

cfg = O.Config(**dict(O.PRESETS["base"], im_size=512, num_channels=1, attn_operands=a.operands))
O.FLASH_FILL_RULE = False
w = O.make_weights(cfg, seed=0)
x, _ = O.make_batch(cfg, B=1, seed=1234)
taps = {}
with torch.no_grad():
    O.forward({k: v.clone() for k, v in w.items()}, O.Config(**dict(cfg.__dict__, attn_drop=0.0, proj_drop=0.0)), x, training=True,
              seed=777, taps=taps)
pre, xin, lvl, _ = [t for t in taps["blocks"] if t[0] == a.block][0]
N, D, hid, s = cfg.level(lvl)
H, Cn, d = cfg.num_heads, cfg.num_channels, D // cfg.num_heads
B = 1
G = torch.randn(B, N, D, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).float()
pp = pre + "ReAttn."
print(f"{pre} level {lvl}: N={N} D={D} d={d}; x mean {xin.mean():.3g} std {xin.std():.3g}")

# ---- oracle with retained intermediate gradients ----
st = torch.bfloat16
p = {k[len(pp):]: w[k].clone() for k in w if k.startswith(pp)}
for k in p:
    if p[k].dtype.is_floating_point and "running" not in k:
        p[k].requires_grad_(True)
xr = xin.to(st).float().requires_grad_(True)
q0 = O._e4(O._r(O.conv3x3_per_patch(xr, Cn, p["qconv2d.weight"]), st), a.operands)
k0 = O._e4(O._r(O.conv3x3_per_patch(xr, Cn, p["kconv2d.weight"]), st), a.operands)
v0 = O._e4(O._r(O.conv3x3_per_patch(xr, Cn, p["vconv2d.weight"]), st), a.operands)
for t in (q0, k0, v0):
    t.retain_grad()
q = q0.reshape(B, N, H, d).permute(0, 2, 1, 3)
k = k0.reshape(B, N, H, d).permute(0, 2, 1, 3)
v = v0.reshape(B, N, H, d).permute(0, 2, 1, 3)
sc = torch.matmul(q, k.transpose(-2, -1)) * d ** -0.5
pr = torch.softmax(sc, -1)
am = O._dropout_quad(pr, a.drop, True, 777, 2 * 3) if a.flash else O._dropout(pr, a.drop, True, 777, 2 * 3, row_pad=8)
W = p["reatten_matrix.weight"].reshape(H, H)
am = torch.einsum("gh,bhij->bgij", W, am) + p["reatten_matrix.bias"].reshape(1, H, 1, 1)
mean, var = am.mean(dim=(0, 2, 3)), am.var(dim=(0, 2, 3), unbiased=False)
ah = (am - mean.reshape(1, H, 1, 1)) * torch.rsqrt(var.reshape(1, H, 1, 1) + 1e-5)
ah = O._r(ah * p["var_norm.weight"].reshape(1, H, 1, 1) + p["var_norm.bias"].reshape(1, H, 1, 1), st)
o = O._r(torch.matmul(ah, v).transpose(1, 2).reshape(B, N, D), st)
o.retain_grad()
y = O._r(F.linear(o, O._r(p["proj.weight"], st), p["proj.bias"]), st)
(y * G).sum().backward()
print(f"oracle: |q| max {q0.abs().max():.4g}, logits max {sc.abs().max():.4g}, rows with max P > 0.999: {(pr.max(-1).values > 0.999).float().mean():.3f}")

# ---- HIP stand-alone op ----
_lib.set_attn_form(a.flash, 0)
L = lib()
dev = "cuda"
KEYS = ["reatten_matrix.weight", "reatten_matrix.bias", "var_norm.weight", "var_norm.bias", "qconv2d.weight", "kconv2d.weight",
        "vconv2d.weight", "proj.weight", "proj.bias"]
dd = {k_: w[pp + k_].to(dev).contiguous() for k_ in KEYS + ["var_norm.running_mean", "var_norm.running_var"]}
pw = w[pp + "proj.weight"].to(st).to(dev).contiguous()
prm = _lib.vu_attn_params(*[dd[k_].data_ptr() for k_ in KEYS[:7]], pw.data_ptr(), dd["proj.bias"].data_ptr(),
                          dd["var_norm.running_mean"].data_ptr(), dd["var_norm.running_var"].data_ptr(), _lib.operand_code(a.operands))
xd, dyd = xin.to(st).to(dev).contiguous(), G.to(st).to(dev).contiguous()
nbytes = L.vu_attn_workspace_bytes(1, B, N, D, H)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
yd, dxd = torch.empty_like(xd), torch.empty_like(xd)
stp = _lib.stream_ptr()
check(L.vu_attn_forward(1, C.byref(prm), ptr(xd), ptr(xd), ptr(yd), None, ptr(ws), nbytes, B, N, D, H, Cn, a.drop, 0.0, 1, 777, 3, stp))
grads = [torch.zeros_like(dd[k_]) for k_ in KEYS]
gs = _lib.vu_attn_grads(*[g.data_ptr() for g in grads])
check(L.vu_attn_backward(1, C.byref(prm), C.byref(gs), ptr(xd), ptr(xd), ptr(dyd), ptr(dxd), None, ptr(ws), nbytes, B, N, D, H, Cn,
                         a.drop, 0.0, 1, 777, 3, stp))
torch.cuda.synchronize()


def al(o_):
    return (o_ + 255) // 256 * 256


# workspace layout of the stand-alone op (csrc/vu_model.hip: carve_attn + carve_attn_ws, bf16): q k v O | Ps Ah | stats lse2 rinv
# delta | pk | dO dq dk dv ...
act, ld = B * N * D * 2, (N + 7) // 8 * 8
mp = B * H * N * ld * 2
off, tens = 0, {}
for name, nb in [("q", act), ("k", act), ("v", act), ("O", act), ("Ps", mp), ("Ah", mp), ("stats", (4 * H * H + 10 * H) * 4),
                 ("lse2", B * H * N * 4), ("rinv", B * H * N * 4), ("delta", B * H * N * 4), ("pk", B * N * D * 4),
                 ("dO", act), ("dq", act), ("dk", act), ("dv", act)]:
    off = al(off)
    tens[name] = (off, nb)
    off += nb


def grab(name, dt=torch.bfloat16):
    o_, nb = tens[name]
    return ws[o_:o_ + nb].view(dt).float().cpu()


def serr(got, ref):
    return ((got.double() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30)).item()


def rel_l2(got, ref):
    return ((got.double() - ref.double()).norm() / (ref.double().norm() + 1e-30)).item()


print(f"y err {serr(yd.float().cpu(), y.detach()):.3e}")
for nm, ref in (("q", q0), ("k", k0), ("v", v0), ("O", o)):
    print(f"  {nm}: scaled max err {serr(grab(nm).reshape(B, N, D), ref.detach()):.3e}")
print(f"  dO: {serr(grab('dO').reshape(B, N, D), o.grad):.3e}")
for nm, ref in (("dq", q0.grad), ("dk", k0.grad), ("dv", v0.grad)):
    got = grab(nm).reshape(B, N, D)
    e = (got - ref).reshape(N, H, d)
    print(f"  {nm}: scaled max err {serr(got, ref):.3e}  relative L2 {rel_l2(got, ref):.3e}  |ref| max {ref.abs().max():.3e} rms {ref.pow(2).mean().sqrt():.3e}"
          f"  error: mean over tokens / rms = {(e.mean(0).abs().max() / (e.pow(2).mean().sqrt() + 1e-30)).item():.3e}"
          f"  sum over tokens got {got.sum(1).abs().max():.3e} ref {ref.sum(1).abs().max():.3e}")
for k_, g in zip(KEYS, grads):
    if k_ != "reatten_matrix.bias":
        print(f"  grad {k_}: {serr(g.cpu(), p[k_].grad):.3e}")
print(f"  dx: {serr(dxd.float().cpu(), xr.grad):.3e}")
