import os, sys, ctypes as C, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd"))
from vit_unet.torch import _lib
from vit_unet.torch._lib import lib, ptr, check
L = lib()
def run(B, N, Cn, s, H, dt, training):
    D = Cn * s * s
    g = torch.Generator().manual_seed(1)
    p = {"mw": (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)), "mb": 0.05 * torch.randn(H, generator=g),
         "bw": 1 + 0.2 * torch.randn(H, generator=g), "bb": 0.1 * torch.randn(H, generator=g),
         "wq": torch.randn(Cn, Cn, 3, 3, generator=g) / (9 * Cn) ** 0.5, "wk": torch.randn(Cn, Cn, 3, 3, generator=g) / (9 * Cn) ** 0.5,
         "wv": torch.randn(Cn, Cn, 3, 3, generator=g) / (9 * Cn) ** 0.5, "pw": torch.randn(D, D, generator=g) / D ** 0.5,
         "pb": 0.05 * torch.randn(D, generator=g), "rm": 0.01 * torch.randn(H, generator=g), "rv": 1e-4 * (1 + torch.rand(H, generator=g))}
    x = torch.randn(B, N, D, generator=g)
    outs = []
    for nolong in (False, True):
        if nolong: os.environ["VU_NO_LONG"] = "1"
        else: os.environ.pop("VU_NO_LONG", None)
        d = {k: v.cuda().contiguous() for k, v in p.items()}
        pw = d["pw"].to(dt).contiguous()
        prm = _lib.vu_attn_params(d["mw"].data_ptr(), d["mb"].data_ptr(), d["bw"].data_ptr(), d["bb"].data_ptr(), d["wq"].data_ptr(),
                                  d["wk"].data_ptr(), d["wv"].data_ptr(), pw.data_ptr(), d["pb"].data_ptr(), d["rm"].data_ptr(), d["rv"].data_ptr())
        xd = x.cuda().to(dt).contiguous()
        code = _lib.DTYPE_CODE[dt]
        nb = L.vu_attn_workspace_bytes(code, B, N, D, H)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        y = torch.empty_like(xd); amap = torch.empty(B, H, N, N, dtype=dt, device="cuda")
        check(L.vu_attn_forward(code, C.byref(prm), ptr(xd), ptr(xd), ptr(y), ptr(amap), ptr(ws), nb, B, N, D, H, Cn, 0.0, 0.0, int(training), 5, 1, _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append((y.float().cpu(), amap.float().cpu()))
    (y0, a0), (y1, a1) = outs
    dm = (a0 - a1).abs()
    print(f"B{B} N{N} D{D} H{H} {dt} train={training}: map max|diff| {dm.max():.3e} (scale {a1.abs().max():.3e}), y max|diff| {(y0-y1).abs().max():.3e} (scale {y1.abs().max():.3e})")
    if dm.max() > 1e-3 * a1.abs().max():
        idx = (dm > 1e-3 * a1.abs().max()).nonzero()
        print("   bad count", idx.shape[0], "first", idx[:5].tolist(), "last", idx[-3:].tolist())
        rows = idx[:, 2].unique(); cols = idx[:, 3].unique()
        print("   bad rows range", rows.min().item(), rows.max().item(), len(rows), " cols range", cols.min().item(), cols.max().item(), len(cols))
for args in [(1, 4096, 1, 8, 8, torch.float32, False), (1, 1024, 1, 16, 8, torch.float32, False), (1, 4096, 1, 8, 8, torch.float32, True),
             (1, 2048, 1, 8, 8, torch.float32, False), (2, 1024, 1, 8, 8, torch.float32, False)]:
    run(*args)
