#!/usr/bin/env python3
"""How far is the library's own GEMM path (csrc/vu_bgemm.hip for the plain big products, csrc/vu_gemm.h otherwise) from the
vendor library on the step's big plain GEMMs?  torch.matmul (hipBLASLt / rocBLAS underneath) against vu_gemm on the same shapes
and layouts; VU_BGEMM=0 times the general 128 x 128 kernel instead.   python tools/gemm_lib_compare.py [--images 64]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-unet_amd"))
import torch  # noqa: E402
from vit_unet.torch import _lib  # noqa: E402
from vit_unet.torch._lib import check, lib, ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, nargs="*", default=[64, 32, 16])
args = ap.parse_args()
L = lib()
dev = "cuda"


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in args.images:
    for M, N, K in ((49 * B, 3072, 3072), (196 * B, 768, 768)):
        dt = torch.bfloat16
        x = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt)          # Linear weight (out, in)
        dy = torch.randn(M, N, device=dev, dtype=dt)
        y = torch.empty(M, N, device=dev, dtype=dt)
        dx = torch.empty(M, K, device=dev, dtype=dt)
        dw = torch.zeros(N, K, device=dev, dtype=torch.float32)
        st = _lib.stream_ptr()
        fl = 2.0 * M * N * K
        # forward y = x w^T; data gradient dx = dy w; weight gradient dw += dy^T x (fp32 accumulate)
        t_f = timed(lambda: check(L.vu_gemm(1, 0, ptr(x), ptr(w), ptr(y), M, N, K, K, 1, 1, K, N, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, st)))
        t_d = timed(lambda: check(L.vu_gemm(1, 0, ptr(dy), ptr(w), ptr(dx), M, K, N, N, 1, K, 1, K, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, st)))
        t_w = timed(lambda: check(L.vu_gemm(1, 1, ptr(dy), ptr(x), ptr(dw), N, K, M, 1, N, K, 1, K, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 1, st)))
        l_f = timed(lambda: torch.matmul(x, w.t(), out=y))
        l_d = timed(lambda: torch.matmul(dy, w, out=dx))
        dwb = torch.empty(N, K, device=dev, dtype=dt)
        l_w = timed(lambda: torch.matmul(dy.t(), x, out=dwb))
        print(f"{B:3d} images M{M} N{N} K{K}: forward vu {t_f:6.1f} us ({fl / t_f / 1e6:6.0f} TF) lib {l_f:6.1f} ({fl / l_f / 1e6:6.0f} TF) | dgrad vu {t_d:6.1f} "
              f"({fl / t_d / 1e6:6.0f}) lib {l_d:6.1f} ({fl / l_d / 1e6:6.0f}) | wgrad vu {t_w:6.1f} ({fl / t_w / 1e6:6.0f}) lib(bf16 out) {l_w:6.1f} ({fl / l_w / 1e6:6.0f})",
              flush=True)
