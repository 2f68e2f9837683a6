#!/usr/bin/env python3
"""Per-kernel times of the non-materialising re-attention (csrc/vu_flash.hip) through the stand-alone op
(vu_attn_forward / vu_attn_backward with VU_ATTN_FLASH=1), measured with the in-library launch profiler.
    python tools/flash_bench.py [--B 64 --N 784 --C 3 --s 8 --H 8 --drop 0.2 --reps 5]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd"))
os.environ["VU_ATTN_FLASH"] = os.environ.get("VU_ATTN_FLASH", "1")
import torch  # noqa: E402
from vit_unet.torch import _lib  # noqa: E402
from vit_unet.torch._lib import check, lib, ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--N", type=int, default=784)
ap.add_argument("--C", type=int, default=3)
ap.add_argument("--s", type=int, default=8)
ap.add_argument("--H", type=int, default=8)
ap.add_argument("--drop", type=float, default=0.2)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--wall", action="store_true", help="event-time forward+backward as a whole, eager and as a hipGraph replay, instead "
                "of the per-kernel profile (the profiler serialises the optional stream fork of the backward)")
a = ap.parse_args()
dev = "cuda"
D = a.C * a.s * a.s
g = torch.Generator().manual_seed(0)
names = ["mix_w", "mix_b", "bn_w", "bn_b", "wq", "wk", "wv", "proj_w", "proj_b"]
p = {"mix_w": torch.eye(a.H) + 0.3 * torch.randn(a.H, a.H, generator=g), "mix_b": 0.05 * torch.randn(a.H, generator=g),
     "bn_w": 1 + 0.2 * torch.randn(a.H, generator=g), "bn_b": 0.1 * torch.randn(a.H, generator=g),
     "wq": torch.randn(a.C, a.C, 3, 3, generator=g) / (9 * a.C) ** 0.5, "wk": torch.randn(a.C, a.C, 3, 3, generator=g) / (9 * a.C) ** 0.5,
     "wv": torch.randn(a.C, a.C, 3, 3, generator=g) / (9 * a.C) ** 0.5, "proj_w": torch.randn(D, D, generator=g) / D ** 0.5,
     "proj_b": 0.05 * torch.randn(D, generator=g)}
d = {k: v.to(dev).contiguous() for k, v in p.items()}
pw = d["proj_w"].to(torch.bfloat16).contiguous()
rm, rv = torch.zeros(a.H, device=dev), torch.ones(a.H, device=dev)
prm = _lib.vu_attn_params(*[d[k].data_ptr() for k in names[:7]], pw.data_ptr(), d["proj_b"].data_ptr(), rm.data_ptr(), rv.data_ptr())
grads = [torch.zeros_like(d[k]) for k in names]
gs = _lib.vu_attn_grads(*[t.data_ptr() for t in grads])
x = torch.randn(a.B, a.N, D, generator=g).to(torch.bfloat16).to(dev)
dy = torch.randn(a.B, a.N, D, generator=g).to(torch.bfloat16).to(dev)
L = lib()
nbytes = L.vu_attn_workspace_bytes(1, a.B, a.N, D, a.H)
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
y, dx = torch.empty_like(x), torch.empty_like(x)
st = _lib.stream_ptr()


def run():
    check(L.vu_attn_forward(1, C.byref(prm), ptr(x), ptr(x), ptr(y), None, ptr(ws), nbytes, a.B, a.N, D, a.H, a.C, a.drop, a.drop, 1, 7, 3, st))
    check(L.vu_attn_backward(1, C.byref(prm), C.byref(gs), ptr(x), ptr(x), ptr(dy), ptr(dx), None, ptr(ws), nbytes, a.B, a.N, D, a.H,
                             a.C, a.drop, a.drop, 1, 7, 3, st))


run()
torch.cuda.synchronize()
if a.wall:
    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    n = max(a.reps, 20)
    eager = timed(run, n)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        st = _lib.stream_ptr()
        run()
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            st = _lib.stream_ptr()
            run()
    torch.cuda.synchronize()
    graph = timed(gr.replay, n)
    print(f"wall: eager {eager:.1f} us, hipGraph replay {graph:.1f} us per fwd+bwd  (B={a.B} N={a.N} fork={os.environ.get('VU_FLASH_FORK', 'auto')})")
    sys.exit(0)
L.vu_prof_enable(C.c_void_p(torch.cuda.current_stream().cuda_stream))
for _ in range(a.reps):
    run()
torch.cuda.synchronize()
rep = json.loads(L.vu_prof_report().decode())
tot = 0.0
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    us = v["ms"] / v["count"] * 1e3
    tot += v["ms"] / a.reps
    print(f"{k:44s} {us:9.1f} us/launch  x{v['count'] // a.reps}")
print(f"total {tot * 1e3:.1f} us per fwd+bwd   (B={a.B} N={a.N} D={D} H={a.H} drop={a.drop} flash={os.environ['VU_ATTN_FLASH']})")
