"""GEMM shapes of the 16 / 32-images-per-GPU steps (BASELINE configs 3-4) under the tile-selection switches:
   python tools/gemm_small_batch.py ; VU_GEMM_TILE=12864 ... ; VU_GEMM_QUARTER_BELOW=100 ... ; VU_GEMM_HALF_BELOW=0 ..."""
import sys, os, torch
sys.path.insert(0, "/root/repo/vit-unet_amd")
from vit_unet.torch import _lib
from vit_unet.torch._lib import lib, ptr, check
L = lib(); dev = "cuda"
def run(M, N, K, form, iters=50):
    dt = torch.bfloat16
    if form == "NN":
        A = torch.randn(M, K, device=dev, dtype=dt); B = torch.randn(N, K, device=dev, dtype=dt); sAm, sAk, sBk, sBn = K, 1, 1, K
    else:
        A = torch.randn(M, K, device=dev, dtype=dt); B = torch.randn(K, N, device=dev, dtype=dt); sAm, sAk, sBk, sBn = K, 1, N, 1
    C = torch.zeros(M, N, device=dev, dtype=dt)
    st = _lib.stream_ptr()
    def call(): check(L.vu_gemm(1, 0, ptr(A), ptr(B), ptr(C), M, N, K, sAm, sAk, sBk, sBn, N, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, st))
    for _ in range(10): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{form} M{M} N{N} K{K}: {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:7.1f} TF tile={os.environ.get('VU_GEMM_TILE','auto')} q={os.environ.get('VU_GEMM_QUARTER_BELOW','200')}")
for args in [(784,3072,3072,"NN"),(784,3072,3072,"NT"),(1568,3072,3072,"NN"),(3136,768,768,"NN"),(3136,768,3072,"NN"),(3136,3072,768,"NN")]:
    run(*args)
