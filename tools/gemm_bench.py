import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-unet_amd"))
from vit_unet.torch import _lib
from vit_unet.torch._lib import lib, ptr, check
L = lib()
dev = "cuda"
def run(M, N, K, form, c_float, acc, iters=20):
    dt = torch.bfloat16
    if form == "TT":
        A = torch.randn(K, M, device=dev, dtype=dt); B = torch.randn(K, N, device=dev, dtype=dt)
        sAm, sAk, sBk, sBn = 1, M, N, 1
    elif form == "NN":
        A = torch.randn(M, K, device=dev, dtype=dt); B = torch.randn(N, K, device=dev, dtype=dt)
        sAm, sAk, sBk, sBn = K, 1, 1, K
    else:
        A = torch.randn(M, K, device=dev, dtype=dt); B = torch.randn(K, N, device=dev, dtype=dt)
        sAm, sAk, sBk, sBn = K, 1, N, 1
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if c_float else dt)
    st = _lib.stream_ptr()
    def call():
        check(L.vu_gemm(1, c_float, ptr(A), ptr(B), ptr(C), M, N, K, sAm, sAk, sBk, sBn, N, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, acc, st))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{form} M{M} N{N} K{K} cf{c_float} acc{acc}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  bk32={os.environ.get('VU_GEMM_BK32','0')}")
for args in [(12544, 768, 64, "NN", 0, 0), (12544, 64, 768, "NN", 0, 0), (50176, 192, 32, "NN", 0, 0), (50176, 32, 192, "NN", 0, 0),
             (3136, 3072, 128, "NN", 0, 0), (3136, 128, 3072, "NN", 0, 0), (12544, 768, 64, "NT", 0, 0), (50176, 192, 192, "NN", 0, 0),
             (12544, 768, 768, "NT", 0, 0), (192, 192, 50176, "TT", 1, 1), (768, 64, 12544, "TT", 1, 1),
             (3072, 3072, 3136, "TT", 1, 1), (768, 768, 12544, "TT", 1, 1), (3072, 3072, 3136, "TT", 0, 0),
             (3136, 3072, 3072, "NN", 0, 0), (3136, 3072, 3072, "NT", 0, 0), (12544, 768, 768, "NN", 0, 0), (50176, 192, 192, "NT", 0, 0)]:
    run(*args)
