"""World-size-2 data-parallel test on CPU (gloo): the bucketed gradient exchange of
vit_unet.torch.engine over the C-side arena layout equals the serial average of per-shard
gradients, bucket ranges tile the arena in backward order, and every rank ends with identical
parameters after the (oracle) AdamW step.  The per-shard gradients come from the CPU oracle (test
infrastructure); BatchNorm statistics stay per replica (spec decision D7)."""
import ctypes as C
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vit_unet_oracle as O
from vit_unet.torch import _lib
from vit_unet.torch.engine import allreduce_bucket, dp_buckets, dp_unit_buckets

KW = dict(depth=1, depth_te=1, size_bottleneck=1, preprocessing="conv", im_size=32, patch_size=8, num_channels=3,
          hidden_dim=16, num_heads=2, attn_drop=0.0, proj_drop=0.0, linear_drop=0.0)


def _flat_grads(cfg, table, total, w, x, y):
    wr = {k: v.clone() for k, v in w.items()}
    for k, _ in O.param_shapes(cfg):
        wr[k].requires_grad_(True)
    O.mse_loss(O.forward(wr, cfg, x, training=True, seed=1), y).backward()
    flat = torch.zeros(total)
    for name, off, shape, _ in table:
        g = wr[name].grad.reshape(-1)
        flat[off:off + g.numel()] = g
    return flat


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = O.Config(**KW)
    ccfg = _lib.make_config(dtype=torch.float32, **KW)
    table = _lib.param_table(ccfg)
    total = _lib.lib().vu_model_param_elems(C.byref(ccfg))
    w = O.make_weights(cfg, seed=0)
    x, y = O.make_batch(cfg, B=4, seed=1234)
    shard = slice(rank * 2, rank * 2 + 2)
    flat = _flat_grads(cfg, table, total, w, x[shard], y[shard])
    # the engine's schedule: buckets of backward units in reverse execution order, every range all-reduced once
    for _first, _last, ranges in dp_unit_buckets(_lib.backward_unit_ranges(ccfg), cap_bytes=64 << 10):
        for lo, hi in ranges:
            allreduce_bucket(flat, lo, hi)
    flat /= world
    ref = sum(_flat_grads(cfg, table, total, w, x[r * 2:r * 2 + 2], y[r * 2:r * 2 + 2]) for r in range(world)) / world
    ok = torch.allclose(flat, ref, rtol=1e-5, atol=1e-7)
    # the bf16 wire option: same schedule, buckets cast to bf16 for the collective: within bf16 rounding of the fp32 result
    flat16 = _flat_grads(cfg, table, total, w, x[shard], y[shard])
    for _first, _last, ranges in dp_unit_buckets(_lib.backward_unit_ranges(ccfg), cap_bytes=64 << 10):
        for lo, hi in ranges:
            allreduce_bucket(flat16, lo, hi, wire_dtype=torch.bfloat16)
    flat16 /= world
    ok = ok and bool(((flat16 - ref).abs() <= 2.0 ** -7 * ref.abs().max()).all()) and not torch.equal(flat16, flat)
    g16 = [torch.zeros_like(flat16) for _ in range(world)]
    dist.all_gather(g16, flat16)
    ok = ok and all(torch.equal(g16[0], g) for g in g16)
    # reduce-scatter + all-gather per bucket (SURVEY 8e's collective for the fully connected xGMI mesh): the same sums as the
    # all-reduce schedule (two ranks: one fp32 addition per element either way - bit-identical), identical on every rank; odd
    # bucket lengths take the small all-reduce for the remainder
    flat_rs = _flat_grads(cfg, table, total, w, x[shard], y[shard])
    for _first, _last, ranges in dp_unit_buckets(_lib.backward_unit_ranges(ccfg), cap_bytes=64 << 10):
        for lo, hi in ranges:
            allreduce_bucket(flat_rs, lo, hi, collective="rs_ag")
    odd = torch.arange(7, dtype=torch.float32) + 10.0 * rank
    allreduce_bucket(odd, 0, 7, collective="rs_ag")
    ok = ok and torch.equal(odd, 2 * torch.arange(7, dtype=torch.float32) + 10.0)
    flat_rs /= world
    ok = ok and torch.equal(flat_rs, flat)
    # identical update on every rank
    p = torch.cat([w[n].reshape(-1) for n, *_ in table])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    ret[rank] = (ok, same)
    dist.destroy_process_group()


def test_bucket_ranges_tile_the_arena():
    for name in ("lite", "base", "large"):
        ccfg = _lib.make_config(dtype=torch.float32, **O.PRESETS[name])
        table = _lib.param_table(ccfg)
        total = _lib.lib().vu_model_param_elems(C.byref(ccfg))
        b = dp_buckets(table, total)
        assert b[0][1] == total and b[-1][0] == 0
        assert b[0][0] == b[1][1] and b[1][0] == b[2][1]          # contiguous, backward order
        names = {n: o for n, o, *_ in table}
        assert b[0][0] == names["Decoders.0.ReAttn.reatten_matrix.weight"]
        assert b[1][0] == names["BottleNeck.0.ReAttn.reatten_matrix.weight"]
        assert all(hi > lo for lo, hi in b)


def test_dp_allreduce_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] == (True, True) and ret[1] == (True, True), dict(ret)


def test_unit_buckets_tile_the_arena_in_backward_order():
    """Buckets built from the C side's backward units cover every gradient exactly once, follow reverse execution
    order, respect the size cap (one unit may exceed it on its own), and leave only first-encoder + PE for the tail."""
    for name in ("base", "large", "lite"):
        ccfg = _lib.make_config(dtype=torch.bfloat16, **O.PRESETS[name])
        units = _lib.backward_unit_ranges(ccfg)
        total = _lib.lib().vu_model_param_elems(C.byref(ccfg))
        table = _lib.param_table(ccfg)
        assert sum(hi - lo for lo, hi in units) == total
        names = {off: n for n, off, *_ in table}
        assert names[units[0][0]].startswith("conv2d.") and units[-1] == (0, table[1][1])      # head first, PE last
        cap = 48 << 20
        buckets = dp_unit_buckets(units, cap)
        covered = sorted(r for _, _, rs in buckets for r in rs)
        assert covered[0][0] == 0 and covered[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))                        # no gap, no overlap
        assert [f for f, _, _ in buckets] == [0] + [l + 1 for _, l, _ in buckets[:-1]]        # consecutive unit ranges
        for f, l, rs in buckets:
            size = sum(hi - lo for lo, hi in rs) * 4
            assert size <= cap or f == l
        f, l, rs = buckets[-1]
        assert names[min(lo for lo, _ in rs)].startswith("PE.")                               # the exposed tail bucket
        assert (3 if name != "lite" else 1) <= len(buckets) <= 12, (name, len(buckets))
        small = dp_unit_buckets(units, 4 << 20)                                               # Lite (20 MB in all) with a 4 MB cap
        assert len(small) >= 3 and sorted(r for _, _, rs in small for r in rs)[0][0] == 0
