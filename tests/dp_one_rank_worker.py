"""Child process of test_dp_path_one_rank_rccl_matches_single_gpu_step (tests/test_gpu_model.py): the data-parallel
engine on one rank over RCCL against the plain single-GPU fused step.  usage: python dp_one_rank_worker.py <golden_dir>
(VU_DP_FORCE=1 in the environment).  Prints DP_ONE_RANK_OK and leaves with os._exit(0): no process-group teardown."""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vit-unet_amd"), os.path.join(ROOT, "oracle"), ROOT):
    sys.path.insert(0, p)
import vit_unet_oracle as O                      # noqa: E402  (test infrastructure: builds the weights)
from vit_unet.torch import model as M            # noqa: E402
from vit_unet.torch.engine import TrainStep      # noqa: E402

DEV = "cuda"


def serr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def build(kw, weights):
    m = M.HViT_UNet(dtype=torch.float32, **kw)
    m.load_state_dict({k: v.clone() for k, v in weights.items()}, strict=True)
    return m.to(DEV)


def main(golden_dir):
    with open(os.path.join(golden_dir, "manifest.json")) as f:
        man = json.load(f)
    g = dict(np.load(os.path.join(golden_dir, "tiny_c.npz")))
    kw = dict(man["cases"]["tiny_c"]["config"], attn_drop=0.2, proj_drop=0.2, linear_drop=0.0)
    w = O.make_weights(O.Config(**kw), seed=7)
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    ma, mb = build(kw, w).train(), build(kw, w).train()
    os.environ.pop("VU_DP_FORCE", None)
    ta = TrainStep(ma, lr=1e-3, seed=5)           # the reference: no process group yet, single-GPU path
    assert not ta.dp
    os.environ["VU_DP_FORCE"] = "1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    tb = TrainStep(mb, lr=1e-3, seed=5, bucket_mb=0)      # cap 0: one bucket (and one all-reduce) per backward unit
    assert tb.dp and tb.world == 1 and tb.comm_stream is not None and len(tb._ubuckets) >= 4
    for _ in range(2):
        la, lb = ta.step(x, y).item(), tb.step(x, y).item()
        assert abs(la - lb) < 1e-4 * abs(la), (la, lb)
    torch.cuda.synchronize()
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        if k.endswith("reatten_matrix.bias"):
            continue
        assert serr(pa, pb) < 1e-4, k
    # per-bucket hipGraphs with the collectives between the graph launches == the eager single-GPU step (fresh models:
    # the comparison is made on the first steps)
    os.environ.pop("VU_DP_FORCE", None)
    mc, md = build(kw, w).train(), build(kw, w).train()
    tc = TrainStep(mc, lr=1e-3, seed=5)
    assert not tc.dp
    os.environ["VU_DP_FORCE"] = "1"
    td = TrainStep(md, lr=1e-3, seed=5, bucket_mb=0)
    td.capture_dp(x, y)                                    # performs one real (eager) warm-up step
    tc.step(x, y)
    for _ in range(2):
        lc, ld = tc.step(x, y).item(), td.replay(x, y).item()
        assert abs(lc - ld) < 2e-4 * abs(lc), (lc, ld)
    torch.cuda.synchronize()
    for (k, pc), (_, pd_) in zip(mc.named_parameters(), md.named_parameters()):
        if not k.endswith("reatten_matrix.bias"):
            assert serr(pc, pd_) < 2e-4, k
    # reduce-scatter + all-gather per bucket over RCCL (collective="rs_ag"): same step as the all-reduce schedule
    me = build(kw, w).train()
    te = TrainStep(me, lr=1e-3, seed=5, bucket_mb=0, collective="rs_ag")
    assert te.dp and te.collective == "rs_ag"
    mf = build(kw, w).train()
    os.environ.pop("VU_DP_FORCE", None)
    tf = TrainStep(mf, lr=1e-3, seed=5)
    assert not tf.dp
    os.environ["VU_DP_FORCE"] = "1"
    for _ in range(2):
        le, lf = te.step(x, y).item(), tf.step(x, y).item()
        assert abs(le - lf) < 1e-4 * abs(lf), (le, lf)
    torch.cuda.synchronize()
    for (k, pe), (_, pf) in zip(me.named_parameters(), mf.named_parameters()):
        if not k.endswith("reatten_matrix.bias"):
            assert serr(pe, pf) < 1e-4, k
    # the library's own RCCL communicator behind the C ABI (include/vit_unet_amd.h: vu_dp_init / vu_dp_allreduce_bucket):
    # created from the process group (the unique id travels through it), one rank: a bucket summed over one rank is unchanged, in
    # fp32 and in bf16, and the step with collective="c_abi" is the single-GPU step
    from vit_unet.torch import _lib
    from vit_unet.torch.engine import dp_c_abi_init
    L = _lib.lib()
    assert L.vu_dp_world() == 0
    assert L.vu_dp_allreduce_bucket(None, 4, 0, None) < 0 and b"vu_dp_init" in L.vu_last_error()      # refuses before vu_dp_init
    assert dp_c_abi_init() == 1 and L.vu_dp_world() == 1 and dp_c_abi_init() == 1
    for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        t = torch.randn(100003, device=DEV).to(dt)
        ref = t.clone()
        _lib.check(L.vu_dp_allreduce_bucket(_lib.ptr(t), t.numel(), code, _lib.stream_ptr(t.device)), "vu_dp_allreduce_bucket")
        torch.cuda.synchronize()
        assert torch.equal(t, ref), dt
    mg, mh = build(kw, w).train(), build(kw, w).train()
    tg = TrainStep(mg, lr=1e-3, seed=5, bucket_mb=0, collective="c_abi")
    assert tg.dp and tg.collective == "c_abi"
    os.environ.pop("VU_DP_FORCE", None)
    th = TrainStep(mh, lr=1e-3, seed=5)
    os.environ["VU_DP_FORCE"] = "1"
    for _ in range(2):
        lg, lh = tg.step(x, y).item(), th.step(x, y).item()
        assert abs(lg - lh) < 1e-4 * abs(lh), (lg, lh)
    torch.cuda.synchronize()
    for (k, pg), (_, ph) in zip(mg.named_parameters(), mh.named_parameters()):
        if not k.endswith("reatten_matrix.bias"):
            assert serr(pg, ph) < 1e-4, k
    assert L.vu_dp_finalize() == 0 and L.vu_dp_world() == 0
    print("DP_ONE_RANK_OK", flush=True)


if __name__ == "__main__":
    try:
        main(sys.argv[1])
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)
