"""Full-size training sanity (not a test: minutes of GPU time): the fused HIP step on a FIXED synthetic batch must bring
the loss down.  python tools/train_sanity.py [base|lite|large|seg512] [steps] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-unet_amd"))
import torch  # noqa: E402
from vit_unet.torch import model as M  # noqa: E402
from vit_unet.torch.engine import TrainStep  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "base"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
torch.manual_seed(0)
g = torch.Generator().manual_seed(1234)
if name == "seg512":
    m = M.get_vit_unet("base", dtype=torch.bfloat16, im_size=512, num_channels=1).to("cuda").train()
    x = torch.rand(B, 1, 512, 512, generator=g).cuda()
    y = (torch.rand(B, 1, 512, 512, generator=g) < 0.1).float().cuda()
    ts = TrainStep(m, lr=1e-3, loss="dice")
else:
    m = M.get_vit_unet(name, dtype=torch.bfloat16).to("cuda").train()
    y = torch.rand(B, 3, 224, 224, generator=g)
    x = (y + 0.1 * torch.randn(y.shape, generator=g)).clamp(0, 1)
    x, y = x.cuda(), y.cuda()
    ts = TrainStep(m, lr=1e-3)
t0 = time.time()
for it in range(steps):
    loss = ts.step(x, y)
    if it % max(1, steps // 10) == 0 or it == steps - 1:
        print(f"{name} step {it:4d} loss {loss.item():.5f}  ({time.time() - t0:.1f} s)", flush=True)
