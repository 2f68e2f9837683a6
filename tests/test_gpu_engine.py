"""Engine / harness state on the GPU: bf16 shadow invalidation, optimizer-state checkpoints, hyper-parameter sync,
BatchNorm counters (round-1 advisor findings)."""
import os

import numpy as np
import pytest
import torch

import vit_unet_oracle as O
from vit_unet.torch import model as M
from vit_unet.torch.engine import TrainStep
from vit_unet.torch.fitter import ImageFitter

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(depth=1, depth_te=1, size_bottleneck=1, preprocessing="conv", im_size=32, patch_size=8, num_channels=3,
          hidden_dim=16, num_heads=2, attn_drop=0.0, proj_drop=0.0, linear_drop=0.0)


def _batch(seed=3, n=4):
    g = torch.Generator().manual_seed(seed)
    y = torch.rand(n, 3, 32, 32, generator=g)
    x = (y + 0.1 * torch.randn(n, 3, 32, 32, generator=g)).clamp(0, 1)
    return x.to(DEV), y.to(DEV)


def test_bf16_shadow_follows_weight_writes_outside_adamw():
    """After a TrainStep exists (it vouches for the bf16 shadow), any torch-side write to a parameter - an in-place
    op, load_state_dict, a torch optimizer step - must reach the next forward."""
    torch.manual_seed(0)
    m = M.HViT_UNet(dtype=torch.bfloat16, **KW).to(DEV)
    x, y = _batch()
    ts = TrainStep(m, lr=1e-3)
    ts.step(x, y)
    m.eval()

    def fresh_out():
        f = M.HViT_UNet(dtype=torch.bfloat16, **KW)
        f.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
        f.to(DEV).eval()
        with torch.no_grad():
            return f(x).clone()
    with torch.no_grad():
        base = m(x).clone()
        assert torch.equal(base, fresh_out())
        for p in m.parameters():                       # (1) in-place write
            p.mul_(0.9)
        o1 = m(x).clone()
        assert not torch.equal(o1, base) and torch.equal(o1, fresh_out())
    sd = {k: (v * 1.05 if v.dtype.is_floating_point else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)                              # (2) load_state_dict
    with torch.no_grad():
        o2 = m(x).clone()
        assert not torch.equal(o2, o1) and torch.equal(o2, fresh_out())
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.05)     # (3) a torch optimizer on the autograd path
    torch.nn.MSELoss()(m(x), y).backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        o3 = m(x).clone()
        assert not torch.equal(o3, o2) and torch.equal(o3, fresh_out())
    m.train()
    l_before = ts.step(x, y).item()                    # the fused step after all that still sees the current weights
    assert np.isfinite(l_before)


def test_fitter_checkpoint_resumes_optimizer_state(tmp_path):
    """save -> load -> step equals the uninterrupted run (AdamW moments and bias-correction step restored);
    weights_only=True restarts the moments (the reference's best-checkpoint reload)."""
    x, y = _batch()
    loader = [{"x": x[:2], "y": y[:2]}, {"x": x[2:], "y": y[2:]}]

    def make():
        torch.manual_seed(1)
        m = M.HViT_UNet(**KW).to(DEV)
        return m, ImageFitter(m, loss=torch.nn.MSELoss(), optimizer=torch.optim.AdamW(m.parameters(), lr=2e-3), device=DEV,
                              folder=str(tmp_path))
    ma, fa = make()
    fa.fit(loader, None, n_epochs=4)                                   # uninterrupted: 4 epochs
    mb, fb = make()
    fb.fit(loader, None, n_epochs=2)
    mc, fc = make()                                                    # "new process": restore and continue
    fc.load(str(tmp_path / "last-checkpoint.bin"))
    assert fc.epoch == 2
    fc.fit(loader, None, n_epochs=2)
    for (k, pa), (_, pc) in zip(ma.named_parameters(), mc.named_parameters()):
        if not k.endswith("reatten_matrix.bias"):
            assert ((pa - pc).abs().max() / (pa.abs().max() + 1e-12)).item() < 1e-4, k
    assert int(fc._fused.step_count.item()) == int(fa._fused.step_count.item()) == 8
    md, fd = make()
    fd.load(str(tmp_path / "last-checkpoint.bin"), weights_only=True)
    fd.fit(loader, None, n_epochs=1)
    assert int(fd._fused.step_count.item()) == 2                       # moments restarted
    # BatchNorm counters follow the fused path too
    assert int(ma.state_dict()["Encoders.0.ReAttn.var_norm.num_batches_tracked"]) == 8


def test_fitter_follows_lr_changes_of_the_users_optimizer(tmp_path):
    x, y = _batch()
    loader = [{"x": x, "y": y}]
    torch.manual_seed(2)
    m = M.HViT_UNet(**KW).to(DEV)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    f = ImageFitter(m, loss=torch.nn.MSELoss(), optimizer=opt, device=DEV, folder=str(tmp_path))
    f.fit(loader, None, n_epochs=1)
    assert abs(f._fused.hyper[0].item() - 1e-3) < 1e-9
    opt.param_groups[0]["lr"] = 0.0                                    # a scheduler would do this
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    f.fit(loader, None, n_epochs=1)
    assert f._fused.hyper[0].item() == 0.0
    for k, p in m.named_parameters():                                  # lr = 0: only weight decay * lr = 0 -> unchanged
        assert torch.equal(p, before[k]), k
    with pytest.raises(NotImplementedError):
        class Crit(torch.nn.Module):
            def forward(self, a, b):
                return (a - b).abs().mean()
        f2 = ImageFitter(m, loss=Crit(), device=DEV, folder=str(tmp_path))
        f2._train_batch(x, y, torch.ones(x.shape[0], device=DEV))


def test_constructor_rejects_what_the_hip_path_cannot_run():
    from vit_unet.torch._lib import VuError
    with pytest.raises(VuError):
        M.HViT_UNet(**dict(KW, num_heads=3, patch_size=12, im_size=36))     # 3 heads: not in {1,2,4,8}
    with pytest.raises(AssertionError):
        M.HViT_UNet(**dict(KW, patch_size=8, depth=2))                       # model.py:281-283 asserts stay asserts
