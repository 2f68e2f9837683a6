"""Fused train step: forward + MSE + backward + (bucketed RCCL all-reduce) + AdamW, all on HIP
kernels over the flat arenas, with optional hipGraph capture.

This is the counterpart of what `benatools.TorchFitterBase.fit` does per batch for
run_denoising.py:78-98 (`out = model(x); loss = MSELoss()(out, y); loss.backward();
AdamW.step()`), minus Python-side autograd: one C call for the forward, one per backward stage,
one for the loss and one for the optimizer.  Data parallelism: one process per GPU, gradients of
the flat arena are all-reduced (sum) in buckets of at most `bucket_mb` (default: a quarter of the
arena clamped to 4-48 MB) cut at backward-unit boundaries in reverse execution order
(`dp_unit_buckets`), each launched on a side stream as soon as its units have been enqueued, so
the collective overlaps the remaining backward; the 1/world average is folded into AdamW.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import VuError, check, lib, ptr, stream_ptr


def dp_buckets(table, total: int):
    """Three contiguous arena ranges in backward order: [Decoders..end], [BottleNeck], [start..Encoders].
    `table` = [(name, offset, shape, bn_index)] from vu_model_param_table."""
    def first(prefix):
        for n, o, *_ in table:
            if n.startswith(prefix):
                return o
        return None
    o_dec = first("Decoders.")
    if o_dec is None:
        o_dec = first("conv2d.")
    if o_dec is None:
        o_dec = total
    o_bot = first("BottleNeck.")
    if o_bot is None:
        o_bot = o_dec
    return [(o_dec, total), (o_bot, o_dec), (0, o_bot)]


def dp_unit_buckets(unit_ranges, cap_bytes: int = 48 << 20, elem_bytes: int = 4):
    """Gradient buckets for the overlapped all-reduce, from the backward UNITS of the C side
    (`vu_model_backward_unit_ranges`: arena range of the gradients each unit produces, in backward = reverse execution
    order: conv2d, SkipConnections.last, Decoders.last ... Encoders.0, PE).  Consecutive units are merged while the bucket
    stays under `cap_bytes` (25-50 MB buckets), so that only the LAST bucket - the first encoder block and the
    positional embedding, whose gradients are complete only when the backward is - cannot overlap with compute.
    Returns [(first_unit, last_unit, [(lo, hi) arena ranges, adjacent ones merged])]."""
    buckets, cur, size, first = [], [], 0, 0
    for u, (lo, hi) in enumerate(unit_ranges):
        nbytes = (hi - lo) * elem_bytes
        if cur and size + nbytes > cap_bytes:
            buckets.append((first, u - 1, cur))
            cur, size, first = [], 0, u
        if hi > lo:
            cur.append((lo, hi))
        size += nbytes
    buckets.append((first, len(unit_ranges) - 1, cur))
    out = []
    for f, l, rs in buckets:
        merged = []
        for lo, hi in sorted(rs):
            if merged and merged[-1][1] == lo:
                merged[-1] = (merged[-1][0], hi)
            else:
                merged.append((lo, hi))
        out.append((f, l, merged))
    return out


def _sum_over_ranks(t: torch.Tensor, group, collective: str):
    """In-place sum of a 1-D tensor over the group.  "all_reduce": one ring all-reduce (per-link bound on xGMI: 2 (w-1)/w S
    through one ~153 GB/s link).  "rs_ag": reduce-scatter + all-gather (on the fully connected xGMI mesh every
    rank exchanges S/w with each of its w-1 peers concurrently in both phases, so the exposed time is ~2 S / (w link_bw)
    instead of ~2 S / link_bw); the same sum, the same value on every rank.  The part of the tensor that does not divide by the
    world size (< w elements) goes through a small all-reduce."""
    if collective == "all_reduce":
        torch.distributed.all_reduce(t, group=group)
        return
    if collective == "c_abi":       # the library's own RCCL communicator (include/vit_unet_amd.h: vu_dp_allreduce_bucket), on the current stream
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise VuError(f"vu_dp_allreduce_bucket sums fp32 or bf16 buckets, not {t.dtype}")
        # ONE communicator per process: it cannot stand in for a sub-group of the torch process group
        if lib().vu_dp_world() != torch.distributed.get_world_size(group):
            raise VuError(f"collective='c_abi': the library's communicator has {lib().vu_dp_world()} ranks, the group "
                          f"{torch.distributed.get_world_size(group)} (dp_c_abi_init(group) first; sub-groups are not supported)")
        check(lib().vu_dp_allreduce_bucket(ptr(t), t.numel(), 0 if t.dtype == torch.float32 else 1, stream_ptr(t.device)),
              "vu_dp_allreduce_bucket")
        return
    w = torch.distributed.get_world_size(group)
    n = t.numel()
    main = n - n % w
    if main:
        shard = torch.empty(main // w, dtype=t.dtype, device=t.device)
        torch.distributed.reduce_scatter_tensor(shard, t[:main], group=group)
        torch.distributed.all_gather_into_tensor(t[:main], shard, group=group)
    if main < n:
        torch.distributed.all_reduce(t[main:], group=group)


def allreduce_bucket(flat: torch.Tensor, lo: int, hi: int, group=None, wire_dtype=None, collective: str = "all_reduce"):
    """Sum one bucket of the flat gradient arena over the data-parallel group (RCCL on GPU
    tensors, gloo on CPU tensors in the tests).  The 1/world average is applied by AdamW.
    `wire_dtype=torch.bfloat16`: the bucket crosses the links as bf16 (half the bytes on the per-link-bound xGMI ring):
    cast, all-reduce, cast back into the fp32 arena - every rank ends with the same values (the sum is formed from the
    same bf16 operands everywhere), each gradient carries one more rounding to 8 bits.
    `collective`: "all_reduce" (default) or "rs_ag" (reduce-scatter + all-gather, `_sum_over_ranks`)."""
    if hi <= lo:
        return
    if wire_dtype is None or wire_dtype == flat.dtype:
        _sum_over_ranks(flat[lo:hi], group, collective)
    else:
        wire = flat[lo:hi].to(wire_dtype)
        _sum_over_ranks(wire, group, collective)
        flat[lo:hi].copy_(wire)


def dp_c_abi_init(group=None) -> int:
    """Create the library's RCCL communicator (vu_dp_init) for the ranks of a torch process group: rank 0 draws the unique id
    (vu_dp_unique_id) and the group broadcasts its 128 bytes.  Returns the world size.  Idempotent."""
    L = lib()
    if L.vu_dp_world():
        return L.vu_dp_world()
    rank, world = torch.distributed.get_rank(group), torch.distributed.get_world_size(group)
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        check(L.vu_dp_unique_id(buf), "vu_dp_unique_id")
    box = [bytes(buf)]
    torch.distributed.broadcast_object_list(box, src=0, group=group)
    raw = (C.c_ubyte * 128).from_buffer_copy(box[0])
    # vu_dp_init creates the communicator on the CURRENT HIP device: make that the rank's device (LOCAL_RANK under torchrun) rather
    # than assuming the caller has done so
    if "LOCAL_RANK" in os.environ and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    check(L.vu_dp_init(rank, world, raw), "vu_dp_init")
    return world


class TrainStep:
    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 process_group=None, seed: int = 0, overlap: bool = True, loss: str = "mse", bucket_mb: Optional[int] = None,
                 grad_wire_dtype: Optional[torch.dtype] = None, collective: str = "all_reduce"):
        """`loss`: "mse" (run_denoising.py:80) or "dice" (README.md:91-101 on sigmoid(model output),
        the segmentation configuration of BASELINE config 5).  `grad_wire_dtype=torch.bfloat16`: data-parallel gradient
        buckets are all-reduced in bf16 (allreduce_bucket); default: fp32, as the reference's DDP would.
        `collective="rs_ag"`: every bucket as reduce-scatter + all-gather instead of one all-reduce (`_sum_over_ranks`);
        `collective="c_abi"`: every bucket through the library's own RCCL communicator (`vu_dp_allreduce_bucket`; `dp_c_abi_init`
        creates it from the torch process group: the unique id travels through the group's store)."""
        if grad_wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("grad_wire_dtype must be None, torch.float32 or torch.bfloat16")
        if collective not in ("all_reduce", "rs_ag", "c_abi"):
            raise ValueError("collective must be 'all_reduce', 'rs_ag' (reduce-scatter + all-gather per bucket) or 'c_abi' (vu_dp_allreduce_bucket)")
        self.collective = collective
        self.grad_wire_dtype = None if grad_wire_dtype == torch.float32 else grad_wire_dtype
        if loss not in ("mse", "dice"):
            raise ValueError(f"loss must be 'mse' or 'dice', got {loss!r}")
        self.loss_kind = loss
        model._ensure_flat()
        self.model = model
        dev = model._arena.device
        self.dev = dev
        n = model._arena.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay], dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)      # device-side (graph replay safe)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.partials = torch.zeros(max(2048, lib().vu_dice_partials_floats()), dtype=torch.float32, device=dev)
        self.seed = seed
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # data-parallel path: more than one rank - or VU_DP_FORCE=1 with an initialised process group, which runs the
        # same bucketed all-reduce / side-stream choreography on ONE rank (how the RCCL path is exercised on a 1-GPU box)
        self.dp = self.world > 1 or (bool(os.environ.get("VU_DP_FORCE")) and torch.distributed.is_available()
                                     and torch.distributed.is_initialized())
        self.overlap = overlap and self.dp
        self.comm_stream = torch.cuda.Stream(device=dev) if self.dp else None
        self._graph = None
        self._seg_graphs = None
        self._gx = self._gy = self._gout = self._dout = None
        self._buckets = self._make_buckets()
        self._nunits = lib().vu_model_num_backward_units(C.byref(model._cfg))
        if bucket_mb is None:      # 25-50 MB buckets, at least ~4 of them for the small models
            bucket_mb = int(min(48, max(4, model._arena.numel() * 4 / 4 / 2 ** 20)))
        self._ubuckets = dp_unit_buckets(_lib.backward_unit_ranges(model._cfg), bucket_mb << 20)
        model._shadow_clean = False
        model.refresh_shadow()
        model._shadow_clean = True      # from now on AdamW keeps the bf16 shadow in sync (model.refresh_shadow re-casts
        #                                 if a torch-side write changed a parameter in between)

    def set_lr(self, lr: float):
        self.hyper[0] = lr

    def _make_buckets(self):
        return dp_buckets(self.model._table, self.model._arena.numel())

    # ---- the step ----------------------------------------------------------------------------
    def _enqueue_head(self, x, y, out, dout):
        """zero the gradient arena, forward, loss (+ dL/dout)"""
        m, L = self.model, lib()
        B = x.shape[0]
        st = stream_ptr(self.dev)
        ws = m._workspace(B)
        salt = self.step_count.view(torch.int32)
        m._garena.zero_()
        check(L.vu_model_forward(C.byref(m._cfg), ptr(m._arena), ptr(m._shadow), ptr(m._bn), ptr(x), ptr(out), ptr(ws), ws.numel(),
                                 B, 1, self.seed, ptr(salt), st), "vu_model_forward")
        if self.loss_kind == "mse":
            check(L.vu_mse_loss(ptr(out), ptr(y), ptr(dout), ptr(self.loss), ptr(self.partials), out.numel(), 1.0, st),
                  "vu_mse_loss")
        else:
            check(L.vu_dice_loss(ptr(out), ptr(y), ptr(dout), ptr(self.loss), ptr(self.partials), out.numel(), 1, 1.0,
                                 st), "vu_dice_loss")

    def _enqueue_units(self, dout, first, last):
        m, L = self.model, lib()
        B = dout.shape[0]
        ws = m._workspace(B)
        salt = self.step_count.view(torch.int32)
        check(L.vu_model_backward_units(C.byref(m._cfg), ptr(m._arena), ptr(m._shadow), ptr(m._bn), ptr(m._garena), ptr(dout),
                                        None, ptr(ws), ws.numel(), B, 1, self.seed, ptr(salt), first, last,
                                        stream_ptr(self.dev)), "vu_model_backward_units")

    def _enqueue_adamw(self):
        m = self.model
        check(lib().vu_adamw(ptr(m._arena), ptr(m._garena), ptr(self.m), ptr(self.v), ptr(m._shadow), m._arena.numel(),
                             ptr(self.hyper), ptr(self.step_count), 1.0 / self.world, stream_ptr(self.dev)), "vu_adamw")

    def _reduce_bucket(self, ranges):
        """all-reduce one gradient bucket on the side stream, after everything enqueued so far on the compute stream"""
        cur = torch.cuda.current_stream(self.dev)
        if self.overlap:
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                for lo, hi in ranges:
                    allreduce_bucket(self.model._garena, lo, hi, self.pg, self.grad_wire_dtype, self.collective)
        else:
            for lo, hi in ranges:
                allreduce_bucket(self.model._garena, lo, hi, self.pg, self.grad_wire_dtype, self.collective)

    def _enqueue(self, x, y, out, dout):
        self._enqueue_head(x, y, out, dout)
        if not self.dp:
            self._enqueue_units(dout, 0, self._nunits - 1)
        else:
            # backward in gradient buckets (reverse execution order): the all-reduce of bucket k runs on the side stream
            # under the backward of bucket k+1; only the last bucket (first encoder block + positional embedding) is exposed
            for first, last, ranges in self._ubuckets:
                self._enqueue_units(dout, first, last)
                self._reduce_bucket(ranges)
            if self.overlap:
                torch.cuda.current_stream(self.dev).wait_stream(self.comm_stream)
        self._enqueue_adamw()

    def step(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """One optimisation step on a float32 (B,C,im,im) batch already resident on the GPU.
        Returns the (device) loss scalar of this step."""
        x = x.float().contiguous()
        y = y.float().contiguous()
        if self._gout is None or self._gout.shape != x.shape:
            self._gout = torch.empty_like(x)
            self._dout = torch.empty_like(x)
        self.model.refresh_shadow()           # no-op unless a parameter was written outside the fused AdamW
        self._enqueue(x, y, self._gout, self._dout)
        self.model._gen += 1
        self.model._nbt_pending += 1          # BatchNorm num_batches_tracked, as the autograd path counts it
        return self.loss

    # ---- optimizer state (checkpoint / resume) ----------------------------------------------------
    def state_dict(self):
        """AdamW moments over the flat arena, the step counter and the hyper-parameters: what
        torch.optim.AdamW.state_dict() carries, in arena layout."""
        return {"exp_avg": self.m.detach().cpu(), "exp_avg_sq": self.v.detach().cpu(), "step": int(self.step_count.item()),
                "hyper": self.hyper.detach().cpu(), "numel": self.m.numel()}

    def load_state_dict(self, sd):
        if sd["numel"] != self.m.numel():
            raise ValueError("optimizer state belongs to a different parameter arena")
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        self.step_count.fill_(int(sd["step"]))
        self.hyper.copy_(sd["hyper"])

    def set_hyper(self, lr=None, betas=None, eps=None, weight_decay=None):
        h = self.hyper.tolist()
        new = [h[0] if lr is None else lr, h[1] if betas is None else betas[0], h[2] if betas is None else betas[1],
               h[3] if eps is None else eps, h[4] if weight_decay is None else weight_decay]
        if new != h:
            self.hyper.copy_(torch.tensor(new, dtype=torch.float32))

    # ---- hipGraph capture ----------------------------------------------------------------------
    def prefers_eager(self, batch: int) -> bool:
        """True when `step` (eager launches) beats `capture` + `replay` for this model at `batch` images per GPU: with more
        workgroups than the chip holds at once, the recompute attention's backward runs its dv sweep on a low-priority
        stream in the tails of the dq / dk sweeps (csrc/vu_flash.hip "Tail overlap"); a captured graph cannot carry the
        priority and takes the serial order.  Base at 64 images: 13.1 ms eager against 13.4 ms replayed; the ~420 launches
        of a step cost the host 2 ms, hidden behind the 13 ms of GPU work."""
        return bool(lib().vu_model_prefers_eager(C.byref(self.model._cfg), int(batch)))

    def capture(self, x: torch.Tensor, y: torch.Tensor):
        """Capture one step into a hipGraph (static input buffers); `replay(x, y)` then copies the
        batch into the static buffers and launches the graph.  Single-GPU only: the collective is
        launched eagerly in DP runs."""
        assert not self.dp, "graph capture is used for the single-GPU path"
        self._gx, self._gy = x.float().contiguous().clone(), y.float().contiguous().clone()
        self._gout, self._dout = torch.empty_like(self._gx), torch.empty_like(self._gx)
        self.model._workspace(x.shape[0])
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            self._enqueue(self._gx, self._gy, self._gout, self._dout)     # warm-up (also a real step)
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue(self._gx, self._gy, self._gout, self._dout)
        self._graph = g

    def replay(self, x: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None:
            self._gx.copy_(x)
            self._gy.copy_(y)
        self.model.refresh_shadow()
        if self._seg_graphs is not None:
            # data-parallel: one hipGraph per gradient bucket (forward + loss ride in the first, AdamW is the last); the
            # collectives are launched between the graph launches, on the side stream
            graphs = self._seg_graphs
            for g, (_, _, ranges) in zip(graphs[:-1], self._ubuckets):
                g.replay()
                self._reduce_bucket(ranges)
            if self.overlap:
                torch.cuda.current_stream(self.dev).wait_stream(self.comm_stream)
            graphs[-1].replay()
        else:
            self._graph.replay()
        self.model._nbt_pending += 1
        return self.loss

    def capture_dp(self, x: torch.Tensor, y: torch.Tensor):
        """Data-parallel counterpart of `capture`: the compute between two collectives is captured as one hipGraph
        (launch-bound at the small per-GPU batches of the DP configurations), the RCCL all-reduces stay eager."""
        assert self.dp
        self._gx, self._gy = x.float().contiguous().clone(), y.float().contiguous().clone()
        self._gout, self._dout = torch.empty_like(self._gx), torch.empty_like(self._gx)
        self.model._workspace(x.shape[0])
        self.step(self._gx, self._gy)                        # warm-up (a real step, eager)
        torch.cuda.synchronize(self.dev)
        graphs = []
        for k, (first, last, _) in enumerate(self._ubuckets):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if k == 0:
                    self._enqueue_head(self._gx, self._gy, self._gout, self._dout)
                self._enqueue_units(self._dout, first, last)
            graphs.append(g)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue_adamw()
        graphs.append(g)
        self._seg_graphs = graphs
