"""Two-rank rehearsal of the data-parallel train step on ONE GPU (gloo moves the CUDA buckets; RCCL cannot put two
ranks on one device).  Launch from a shell that has not touched the GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_rehearsal.py

Checks, after 3 fused steps with per-rank batches: (1) every rank holds bit-identical parameters (the three bucketed
all-reduces cover the whole gradient arena and AdamW applies the same 1/world average everywhere); (2) rank 0's parameters
equal a single-process reference that back-propagates both ranks' batches through the autograd path, averages the
gradients and steps torch.optim.AdamW (BatchNorm statistics per replica, as in the DP run).

With `--base`: the Base preset in bf16 at 20 images per rank (level 2 runs the recompute attention form, dropout on), ONE
step; checks (1) as above and (3) the all-reduced gradient arena of the DP step equals, bit for bit, the sum of the two
ranks' gradient arenas recomputed without data parallelism (same seeds; every kernel of the step sums in a fixed order and
the two-rank all-reduce is one commutative fp32 addition per element)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-unet_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from vit_unet.torch import model as M  # noqa: E402
from vit_unet.torch.engine import TrainStep  # noqa: E402

if "--base" in sys.argv:
    BS = 20

    def base_model():
        torch.manual_seed(0)
        return M.get_vit_unet("base", dtype=torch.bfloat16).to("cuda").train()

    def base_batch(r):
        g = torch.Generator().manual_seed(1000 + r)
        y = torch.rand(BS, 3, 224, 224, generator=g)
        return (y + 0.1 * torch.randn(y.shape, generator=g)).clamp(0, 1).cuda(), y.cuda()

    m = base_model()
    ts = TrainStep(m, lr=1e-4, seed=7 + rank)
    assert ts.dp and ts.world == world
    ts.step(*base_batch(rank))
    torch.cuda.synchronize()
    dp_out = ts._gout.detach().clone()
    summed = m._garena.detach().clone()               # after the bucketed all-reduces: the SUM over the ranks (AdamW folds 1 / world in)
    arena = m._arena.detach().clone()
    gathered = [torch.empty_like(arena) for _ in range(world)]
    dist.all_gather(gathered, arena)
    ok_same = all(torch.equal(gathered[0], t) for t in gathered)
    ok_sum, nbad = True, 0
    if rank == 0:
        total = None
        for r in range(world):
            mr = base_model()
            tr = TrainStep(mr, lr=1e-4, seed=7 + r)
            tr.dp, tr.overlap, tr.world = False, False, 1            # the same engine without the collectives
            x, y = base_batch(r)
            out, dout = torch.empty_like(x), torch.empty_like(x)
            tr._enqueue_head(x, y, out, dout)
            tr._enqueue_units(dout, 0, tr._nunits - 1)
            torch.cuda.synchronize()
            if r == 0:       # the forward of rank 0's data-parallel step against the same forward without data parallelism
                nfw = int((out != dp_out).sum().item())
                print(f"  forward output of rank 0: {nfw} of {out.numel()} elements differ from the data-parallel step's "
                      f"(max |diff| {(out - dp_out).abs().max().item():.3e})", flush=True)
            total = mr._garena.detach().clone() if total is None else total + mr._garena
            del mr, tr
        nbad = int((total != summed).sum().item())
        ok_sum = nbad == 0
        if nbad:
            base_ptr = m._arena.data_ptr()
            for name, prm in m.named_parameters():
                o = (prm.data_ptr() - base_ptr) // 4
                d = (total[o:o + prm.numel()] != summed[o:o + prm.numel()])
                if d.any():
                    e = (total[o:o + prm.numel()] - summed[o:o + prm.numel()]).abs().max().item()
                    print(f"  differs: {name} {tuple(prm.shape)}: {int(d.sum())} elements, max |diff| {e:.3e}, max |g| {summed[o:o + prm.numel()].abs().max().item():.3e}", flush=True)
    print(f"rank {rank}: base bf16 / recompute form: identical parameters across ranks: {ok_same}; all-reduced gradients equal the "
          f"sum of the per-rank gradients bit for bit: {ok_sum} ({nbad} of {summed.numel()} differ)", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if (ok_same and ok_sum) else 1)

kw = dict(depth=1, depth_te=1, size_bottleneck=1, preprocessing="conv", im_size=32, patch_size=8, num_channels=3,
          hidden_dim=16, num_heads=2, attn_drop=0.0, proj_drop=0.0, linear_drop=0.0)


def build():
    torch.manual_seed(0)
    return M.HViT_UNet(**kw).to("cuda").train()


def batch(r, step):
    g = torch.Generator().manual_seed(100 * step + r)
    y = torch.rand(4, 3, 32, 32, generator=g)
    return (y + 0.1 * torch.randn(y.shape, generator=g)).clamp(0, 1).cuda(), y.cuda()


m = build()
ts = TrainStep(m, lr=1e-3, seed=7 + rank)
assert ts.dp and ts.world == world
for step in range(3):
    ts.step(*batch(rank, step))
torch.cuda.synchronize()
arena = m._arena.detach().clone()
gathered = [torch.empty_like(arena) for _ in range(world)]
dist.all_gather(gathered, arena)
ok_same = all(torch.equal(gathered[0], t) for t in gathered)
ok_ref, err = True, 0.0
if rank == 0:
    ref = build()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    bn0 = {k: v.clone() for k, v in ref.state_dict().items() if "running" in k or "num_batches" in k}
    for step in range(3):
        grads = None
        for r in range(world):
            if r > 0:    # BatchNorm running statistics are per replica: rank 0's model only sees rank 0's batches
                saved = {k: v.clone() for k, v in ref.state_dict().items() if k in bn0}
            opt.zero_grad()
            x, y = batch(r, step)
            torch.nn.MSELoss()(ref(x), y).backward()
            if r > 0:
                ref.load_state_dict(saved, strict=False)
            gs = [p.grad.detach().clone() for p in ref.parameters()]
            grads = gs if grads is None else [a + b for a, b in zip(grads, gs)]
        for p, gsum in zip(ref.parameters(), grads):
            p.grad = gsum / world
        opt.step()
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if k.endswith("reatten_matrix.bias"):
            continue      # exact gradient 0 under train-mode BatchNorm: Adam turns rounding noise into +-lr steps
        e = ((p.detach() - q.detach()).abs().max() / (q.detach().abs().max() + 1e-30)).item()
        err = max(err, e)
    ok_ref = err < 1e-3      # float-atomic order differs between the fused step and autograd; Adam amplifies it over 3 steps
print(f"rank {rank}: identical parameters across ranks: {ok_same}; vs single-process reference: {ok_ref} (max scaled err {err:.2e})",
      flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if (ok_same and ok_ref) else 1)
