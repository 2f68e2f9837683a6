"""A few launches of the big-tile GEMM (csrc/vu_bgemm.hip) in its three operand forms, for rocprofv3 counter passes.
    python tools/bgemm_probe.py [--images 64] [--reps 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-unet_amd"))
import torch  # noqa: E402
from vit_unet.torch import _lib  # noqa: E402
from vit_unet.torch._lib import check, lib, ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=64)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
L = lib()
M, N, K = 49 * a.images, 3072, 3072
dt = torch.bfloat16
x = torch.randn(M, K, device="cuda", dtype=dt)
w = torch.randn(N, K, device="cuda", dtype=dt)
dy = torch.randn(M, N, device="cuda", dtype=dt)
y = torch.empty(M, N, device="cuda", dtype=dt)
dx = torch.empty(M, K, device="cuda", dtype=dt)
dw = torch.zeros(N, K, device="cuda", dtype=torch.float32)
st = _lib.stream_ptr()
for _ in range(a.reps):
    check(L.vu_gemm(1, 0, ptr(x), ptr(w), ptr(y), M, N, K, K, 1, 1, K, N, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, st))
    check(L.vu_gemm(1, 0, ptr(dy), ptr(w), ptr(dx), M, K, N, N, 1, K, 1, K, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, st))
    check(L.vu_gemm(1, 1, ptr(dy), ptr(x), ptr(dw), N, K, M, 1, N, K, 1, K, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 1, st))
    torch.matmul(x, w.t(), out=y)
torch.cuda.synchronize()
print("done")
