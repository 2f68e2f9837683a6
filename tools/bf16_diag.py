"""Diagnostic (not a test): how far do bf16-storage gradients drift from fp32 ones on full-size
Base, for the HIP path and for the CPU oracle's bf16-storage emulation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import vit_unet_oracle as O
from vit_unet.torch import model as M

def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()

def run(kw, B, tag):
    cfg = O.Config(**kw)
    if os.environ.get("DIAG_INIT", "default") == "default":     # torch default initialisers (what training starts from)
        torch.manual_seed(0)
        w = {k: v.detach().clone() for k, v in M.HViT_UNet(**kw).state_dict().items()}
    else:
        w = O.make_weights(cfg, seed=0)
    x, y = O.make_batch(cfg, B=B, seed=1234)
    names = [k for k, _ in O.param_shapes(cfg)]
    res = {}
    for st in (None, torch.bfloat16):
        wr = {k: v.clone() for k, v in w.items()}
        for k in names: wr[k].requires_grad_(True)
        out = O.forward(wr, cfg, x, training=True, seed=5, storage=st)
        O.mse_loss(out, y).backward()
        res["o32" if st is None else "o16"] = (out.detach(), torch.cat([wr[k].grad.reshape(-1) for k in names]))
    for dt in (torch.float32, torch.bfloat16):
        m = M.HViT_UNet(dtype=dt, **kw)
        m.load_state_dict({k: v.clone() for k, v in w.items()})
        m = m.to("cuda").train(); m._step_seed = 5
        out = m(x.cuda()); torch.nn.MSELoss()(out, y.cuda()).backward()
        sd = dict(m.named_parameters())
        res["h32" if dt == torch.float32 else "h16"] = (out.detach().cpu(), torch.cat([sd[k].grad.reshape(-1).cpu() for k in names]))
    def rel(a, b): return ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()).item()
    print(f"[{tag}] fwd relRMS  h32-o32 {rel(res['h32'][0], res['o32'][0]):.2e}  o16-o32 {rel(res['o16'][0], res['o32'][0]):.2e}  "
          f"h16-o16 {rel(res['h16'][0], res['o16'][0]):.2e}  h16-h32 {rel(res['h16'][0], res['h32'][0]):.2e}")
    print(f"[{tag}] grad cos    h32-o32 {cos(res['h32'][1], res['o32'][1]):.4f}  o16-o32 {cos(res['o16'][1], res['o32'][1]):.4f}  "
          f"h16-o16 {cos(res['h16'][1], res['o16'][1]):.4f}  h16-h32 {cos(res['h16'][1], res['h32'][1]):.4f}", flush=True)

base = dict(O.PRESETS["base"], attn_drop=0.0, proj_drop=0.0)
run(dict(base, depth=0, depth_te=1, size_bottleneck=1), 2, "1 block @L0")
run(dict(base, depth=1, depth_te=1, size_bottleneck=1), 2, "depth1 te1 bot1 (3 blocks + skip)")
run(dict(base, depth=2, depth_te=1, size_bottleneck=1), 2, "depth2 te1 bot1")
run(base, 2, "base B=2")
run(dict(base, attn_drop=0.2, proj_drop=0.2), 4, "base drop B=4")
