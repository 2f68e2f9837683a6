"""Two runs of the same forward + backward (same weights, batch and dropout seed): which parameter gradients are not
bit-identical?  (float atomics anywhere in the backward show up here.)  usage: python tools/determinism_check.py [B]"""
import sys, torch
sys.path.insert(0, "/root/repo/vit-unet_amd"); sys.path.insert(0, "/root/repo/oracle")
import vit_unet.torch.model as M
import vit_unet_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = M.get_vit_unet("base", dtype=torch.bfloat16).cuda().train()
cfg = O.Config(**O.PRESETS["base"])
x, y = O.make_batch(cfg, B=B, seed=5)
x, y = x.cuda(), y.cuda()
runs = []
for rep in range(2):
    m.zero_grad(set_to_none=False)
    m._step_seed = 1234
    out = m(x)
    loss = torch.nn.MSELoss()(out, y)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
print("outputs identical:", torch.equal(runs[0][0], runs[1][0]))
bad = [(k, ((runs[0][1][k] - runs[1][1][k]).abs().max() / (runs[0][1][k].abs().max() + 1e-30)).item())
       for k in runs[0][1] if not torch.equal(runs[0][1][k], runs[1][1][k])]
print(f"B={B}: gradients differing: {len(bad)} of {len(runs[0][1])}")
for k, d in bad[:60]:
    print(f"   {k}  rel {d:.2e}")
