import sys, os, torch
sys.path.insert(0, "/root/repo/vit-unet_amd")
from vit_unet.torch import _lib
from vit_unet.torch._lib import lib, ptr, check
L = lib(); dev = "cuda"
def run(M, N, K, form, addend=False, iters=30):
    dt = torch.bfloat16
    if form == "NN":
        A = torch.randn(M, K, device=dev, dtype=dt); B = torch.randn(N, K, device=dev, dtype=dt); sAm, sAk, sBk, sBn = K, 1, 1, K
    else:
        A = torch.randn(M, K, device=dev, dtype=dt); B = torch.randn(K, N, device=dev, dtype=dt); sAm, sAk, sBk, sBn = K, 1, N, 1
    C = torch.zeros(M, N, device=dev, dtype=dt)
    st = _lib.stream_ptr()
    def call(): check(L.vu_gemm(1, 0, ptr(A), ptr(B), ptr(C), M, N, K, sAm, sAk, sBk, sBn, N, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, st))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{form} M{M} N{N} K{K}: {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:7.1f} TF  {(M*K+K*N+M*N)*2/ms/1e6:7.1f} GB/s tile={os.environ.get('VU_GEMM_TILE','auto')}")
for args in [(50176,192,192,"NN"),(50176,192,192,"NT"),(12544,768,768,"NN"),(50176,32,192,"NN"),(50176,192,32,"NN"),(3136,128,3072,"NN"),(3136,3072,128,"NN"),(12544,64,768,"NN"),(12544,768,64,"NN")]:
    run(*args)
