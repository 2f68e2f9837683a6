"""Per-launch times of the level-2 FeedForward kernels (launch profiler tags) at B x 784 token rows: the fused route of
csrc/vu_ff2.hip (VU_FF2 unset / 1) or the two-launch route (VU_FF2=0).  usage: python tools/ff2_time.py [B = 64]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-unet_amd"))
import torch
from vit_unet.torch import model as M
from vit_unet.torch._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
ff = M.FeedForward(192, 32, 0.0).to(dev).train()
x = torch.randn(B, 784, 192, device=dev).bfloat16().requires_grad_(True)
dy = torch.randn(B, 784, 192, device=dev).bfloat16()
for _ in range(3):
    ff(x).backward(dy)
torch.cuda.synchronize()
L = lib()
L.vu_prof_enable(C.c_void_p(torch.cuda.current_stream().cuda_stream))
for _ in range(20):
    ff(x).backward(dy)
torch.cuda.synchronize()
rep = json.loads(L.vu_prof_report().decode())
for k, v in rep.items():
    print("VU_FF2=%s B=%d %-60s %s" % (os.environ.get("VU_FF2", "-"), B, k[:60], json.dumps(v)[:160]))
