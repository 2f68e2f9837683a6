import sys, torch
sys.path.insert(0, "/root/repo/vit-unet_amd"); sys.path.insert(0, "/root/repo/oracle")
import vit_unet.torch.model as M
from vit_unet.torch.engine import TrainStep
import vit_unet_oracle as O
torch.manual_seed(0)
m0 = M.get_vit_unet("base", dtype=torch.bfloat16)
sd = {k: v.clone() for k, v in m0.state_dict().items()}
cfg = O.Config(**O.PRESETS["base"])
x, y = O.make_batch(cfg, B=8, seed=5)
x, y = x.cuda(), y.cuda()
res = []
for rep in range(2):
    m = M.get_vit_unet("base", dtype=torch.bfloat16); m.load_state_dict(sd); m = m.cuda().train()
    ts = TrainStep(m, lr=1e-3, seed=3)
    for _ in range(3): l = ts.step(x, y)
    torch.cuda.synchronize()
    res.append(({k: p.detach().clone() for k, p in m.named_parameters()}, l.item()))
bad = [(k, (res[0][0][k] - res[1][0][k]).abs().max().item()) for k in res[0][0] if not torch.equal(res[0][0][k], res[1][0][k])]
print("loss", res[0][1], res[1][1], "params differing:", len(bad), "of", len(res[0][0]))
for k, d in bad[:40]: print("  ", k, d)
