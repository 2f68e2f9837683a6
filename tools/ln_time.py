"""Per-call GPU time of the block LayerNorm (K13) forward and backward through the C-ABI at B x 150528 bf16 elements, by the launch
profiler behind a gate kernel (the launches run back to back).  usage: [VU_LIB_PATH=...] python tools/ln_time.py [B = 64] [P = 150528]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-unet_amd"))
import torch
from vit_unet.torch import _lib
from vit_unet.torch._lib import lib, ptr, stream_ptr, check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(sys.argv[2]) if len(sys.argv) > 2 else 150528
dev = torch.device("cuda:0")
L = lib()
a = torch.randn(B, P, device=dev).bfloat16()
dy = torch.randn(B, P, device=dev).bfloat16()
w, b = torch.randn(P, device=dev), torch.randn(P, device=dev)
y, dz = torch.empty_like(a), torch.empty_like(a)
dw, db = torch.zeros_like(w), torch.zeros_like(w)
ws = torch.empty(L.vu_layernorm_workspace_floats(B, P), dtype=torch.float32, device=dev)
stats = torch.empty(B, 2, dtype=torch.float32, device=dev)
sp = stream_ptr(dev)
def fwd(): check(L.vu_add_layernorm_fwd(1, ptr(a), None, ptr(a), ptr(w), ptr(b), ptr(y), ptr(ws), ptr(stats), B, P, sp), "fwd")
def bwd(): check(L.vu_layernorm_bwd(1, ptr(dy), ptr(a), ptr(w), ptr(stats), ptr(dw), ptr(db), ptr(ws), ptr(dz), B, P, sp), "bwd")
for _ in range(3):
    fwd(); bwd()
torch.cuda.synchronize()
L.vu_prof_enable(C.c_void_p(torch.cuda.current_stream().cuda_stream))
check(L.vu_prof_gate(20000), "gate")
for _ in range(30):
    fwd(); bwd()
torch.cuda.synchronize()
rep = json.loads(L.vu_prof_report().decode())
for k, v in rep.items():
    print("lib=%s B=%d P=%d %-26s %.2f us per call" % (os.path.basename(os.environ.get("VU_LIB_PATH", "default")), B, P, k, 1e3 * v["ms"] / v["count"]))
