#!/usr/bin/env python3
"""Back-to-back timing of the fused q/k/v convolution and its data gradient (csrc/vu_conv.hip stencil form against
csrc/vu_conv_mm.hip matrix-core form; VU_CONV_MM=0 selects the former).
    python tools/conv_bench.py [--B 64] [--reps 50]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd"))
import torch  # noqa: E402
from vit_unet.torch import _lib  # noqa: E402
from vit_unet.torch._lib import check, lib, ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
L = lib()
st = _lib.stream_ptr()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C_, s, N in ((3, 8, 784), (3, 16, 196), (3, 32, 49), (1, 8, 4096)):
    npatch = a.B * N
    x = torch.randn(npatch, C_ * s * s, device="cuda").to(torch.bfloat16)
    x2 = torch.randn_like(x)
    w = [torch.randn(C_, C_, 3, 3, device="cuda") * 0.3 for _ in range(3)]
    o = [torch.empty_like(x) for _ in range(5)]
    mb = x.numel() * 2 / 1e6
    f = timed(lambda: check(L.vu_conv3x3_qkv_fwd(1, ptr(x), ptr(x), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(o[0]), ptr(o[1]), ptr(o[2]), npatch, C_, s, st)), a.reps)
    fc = timed(lambda: check(L.vu_conv3x3_qkv_fwd(1, ptr(x), ptr(x2), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(o[0]), ptr(o[1]), ptr(o[2]), npatch, C_, s, st)), a.reps)
    d = timed(lambda: check(L.vu_conv3x3_qkv_dgrad(1, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(x), None, ptr(o[3]), None, npatch, C_, s, st)), a.reps)
    dc = timed(lambda: check(L.vu_conv3x3_qkv_dgrad(1, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(x), ptr(x2), ptr(o[3]), ptr(o[4]), npatch, C_, s, st)), a.reps)
    dw = [torch.zeros(C_, C_, 3, 3, device="cuda") for _ in range(3)]
    scr = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
    wg = timed(lambda: check(L.vu_conv3x3_qkv_wgrad(1, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(x), ptr(x), ptr(dw[0]), ptr(dw[1]), ptr(dw[2]), ptr(scr), scr.numel(), npatch, C_, s, st)), a.reps)
    print(f"C={C_} s={s:2d} npatch={npatch:7d} ({mb:5.1f} MB per tensor)  fwd {f:6.1f} us ({4 * mb / f:5.2f} TB/s)  fwd cross {fc:6.1f}  "
          f"dgrad {d:6.1f} us ({5 * mb / d:5.2f} TB/s)  dgrad cross {dc:6.1f}  wgrad(+reduce) {wg:6.1f} us ({4 * mb / wg:5.2f} TB/s)  tz={os.environ.get('VU_CONV_TZ', '1')}")
