"""Every launch-profiler tag of one train step (count, us per launch, us per step), the instrumented steps behind a gate kernel.
usage: python tools/step_tags.py [--model base] [--batch 64] [--grep substring]"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-unet_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch
import vit_unet_oracle as O
from vit_unet.torch import _lib, model as M
from vit_unet.torch.engine import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="base"); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--grep", default="")
ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = O.Config(**O.PRESETS[a.model])
x, y = O.make_batch(cfg, B=a.batch, seed=3)
torch.manual_seed(0)
m = M.get_vit_unet(a.model, dtype=torch.bfloat16).to(dev).train()
ts = TrainStep(m, lr=1e-3, seed=1)
x, y = x.to(dev), y.to(dev)
for _ in range(3):
    ts.step(x, y)
torch.cuda.synchronize()
L = _lib.lib()
L.vu_prof_enable(C.c_void_p(torch.cuda.current_stream().cuda_stream))
for _ in range(a.steps):
    _lib.check(L.vu_prof_gate(60000), "gate")
    ts.step(x, y)
torch.cuda.synchronize()
rep = json.loads(L.vu_prof_report().decode())
tot = 0.0
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    tot += v["ms"]
    if a.grep in k:
        print("%-6s B=%-3d LAST_BLOCK=%s %-44s n/step %5.1f  %7.2f us/launch  %8.1f us/step" % (a.model, a.batch, os.environ.get("VU_LAST_BLOCK", "-"), k[:44], v["count"] / a.steps, 1e3 * v["ms"] / v["count"], 1e3 * v["ms"] / a.steps))
print("sum of tags: %.1f us/step" % (1e3 * tot / a.steps))
