"""Child of test_c_abi_allreduce_matches_torch_on_two_gpus (tests/test_zz_dp_gpu.py), one process per GPU under torch.distributed.run:
the library's own RCCL communicator (vu_dp_init / vu_dp_allreduce_bucket, csrc/vu_dp.cpp) against torch.distributed.all_reduce on
the same buffers, fp32 and bf16, bit for bit.  Prints C_ABI_OK <rank> and leaves with os._exit(0) (no process-group teardown)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd"))
from vit_unet.torch import engine  # noqa: E402
from vit_unet.torch._lib import lib  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
assert engine.dp_c_abi_init() == world and lib().vu_dp_world() == world
g = torch.Generator().manual_seed(100 + rank)
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(1 << 20, generator=g).to(dt).cuda()
    a, b = x.clone(), x.clone()
    dist.all_reduce(a)
    engine._sum_over_ranks(b, None, "c_abi")
    torch.cuda.synchronize()
    assert torch.equal(a, b), dt
# a bucket of the flat arena through allreduce_bucket, both collectives
flat = torch.randn(3 << 20, generator=g).cuda()
fa, fb = flat.clone(), flat.clone()
engine.allreduce_bucket(fa, 4096, (2 << 20) + 12, collective="all_reduce")
engine.allreduce_bucket(fb, 4096, (2 << 20) + 12, collective="c_abi")
torch.cuda.synchronize()
assert torch.equal(fa, fb)
try:
    engine._sum_over_ranks(torch.zeros(8, dtype=torch.float16, device="cuda"), None, "c_abi")
    raise AssertionError("fp16 must be refused")
except engine.VuError:
    pass
print("C_ABI_OK", rank, flush=True)
dist.barrier()
torch.cuda.synchronize()
sys.stdout.flush()
os._exit(0)
