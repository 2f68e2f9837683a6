"""Which kernel loses run-to-run bit-identity when ANOTHER PROCESS uses the same GPU?  (round 4)

    python tools/contention_ops.py [--load 60] [--iters 300] [--B 20]

The round-3 two-rank rehearsal went red because the Base bf16 forward differed between two runs once a second process was
computing on the same GPU; tools/nondet_check.py --ws-diff named the q / k / v buffers as the first to differ.  This runs the
stand-alone ops at Base level shapes back to back under that load and counts repetitions whose output bytes differ from the
first.  Each op is deterministic by construction (no atomics on these paths), so any count > 0 is a bug in that op."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--load", type=float, default=60.0)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--B", type=int, default=20)
ap.add_argument("--attn", default="h8", choices=["none", "h8", "all"],
                help="also run the recompute attention op forward + backward: h8 = Base level 2 (8 heads), all = + the 4-head Lite levels "
                     "(known, DESIGN 2a: their dq / dk sweeps are NOT reproducible under GPU sharing)")
ap.add_argument("--load-mode", default="process", choices=["process", "thread"],
                help="process: the load is a second PROCESS on the GPU (its queues are time-sliced against ours: waves can be saved / "
                     "restored); thread: the SAME train-step loop on a second stream of THIS process (concurrent kernels, one process: "
                     "no time-slicing between processes).  Round 6, review item 6: do the two sharing faults need a second process?")
ap.add_argument("--only", default="", help="comma list of op families to run: conv,dgrad,wgrad,ln,retile,gemm (default: all of them)")
args = ap.parse_args()
only = set(x for x in args.only.split(",") if x)
child = None
if args.load > 0 and args.load_mode == "process":
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "nondet_check.py"), "--as-load", str(args.load), "--B", str(args.B)])
sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd"))
import torch  # noqa: E402
from vit_unet.torch import _lib  # noqa: E402
from vit_unet.torch._lib import check, lib, ptr  # noqa: E402

L = lib()
dev = "cuda"
bf = torch.bfloat16
if child is not None:
    import glob
    t0 = time.time()
    while not glob.glob(f"/tmp/nondet_load_{child.pid}") and time.time() - t0 < 180:
        time.sleep(0.2)
    print(f"load process running after {time.time() - t0:.1f} s", flush=True)
load_thread = None
if args.load > 0 and args.load_mode == "thread":
    import threading
    from vit_unet.torch import model as M_
    from vit_unet.torch.engine import TrainStep
    started, nsteps = threading.Event(), [0]

    def load_loop():       # what tools/nondet_check.py --as-load does, on a stream of its own inside this process
        torch.cuda.set_device(0)
        s2 = torch.cuda.Stream()
        with torch.cuda.stream(s2):
            torch.manual_seed(0)
            m_ = M_.get_vit_unet("base", dtype=torch.bfloat16).to("cuda").train()
            ts_ = TrainStep(m_, lr=1e-4, seed=7)
            g_ = torch.Generator().manual_seed(1000)
            y_ = torch.rand(args.B, 3, 224, 224, generator=g_)
            x_ = (y_ + 0.1 * torch.randn(y_.shape, generator=g_)).clamp(0, 1).cuda()
            y_ = y_.cuda()
            t0_ = time.time()
            while time.time() - t0_ < args.load:
                ts_.step(x_, y_)
                s2.synchronize()
                nsteps[0] += 1
                started.set()
        print(f"[load thread] {nsteps[0]} steps", flush=True)
    load_thread = threading.Thread(target=load_loop, daemon=True)
    load_thread.start()
    t0 = time.time()
    started.wait(180)
    print(f"load thread running after {time.time() - t0:.1f} s (second stream of this process)", flush=True)
st = _lib.stream_ptr()
B = args.B
g = torch.Generator(device="cpu").manual_seed(3)
total_bad = 0


def conv_detail(s_):
    def show(outs, ref):
        for name_, o, r in zip("qkv", outs, ref):
            d = (o.view(torch.int16) != r.view(torch.int16)).reshape(-1)
            idx = d.nonzero().reshape(-1)
            if idx.numel() == 0:
                continue
            ss = s_ * s_
            print(f"    {name_}: {idx.numel()} elements differ; first 12:", flush=True)
            for i in idx[:12].tolist():
                patch, rem = divmod(i, 3 * ss)
                ch, rem2 = divmod(rem, ss)
                y_, x_ = divmod(rem2, s_)
                quad = (patch * 3 * ss + ch * ss + y_ * s_ + x_) // 4
                print(f"      elem {i}: patch {patch} ch {ch} y {y_} x {x_}  got {o.reshape(-1)[i].item():.5f} ref {r.reshape(-1)[i].item():.5f}", flush=True)
            quads = torch.unique(idx // 4)
            print(f"      {quads.numel()} quads; distinct patches {torch.unique(idx // (3 * ss)).numel()}; channels {torch.unique((idx % (3 * ss)) // ss).tolist()}; "
                  f"runs of consecutive elements: {int((idx[1:] - idx[:-1] != 1).sum()) + 1}", flush=True)
    return show


def repeat(name, fn, outs, detail=None):
    """fn() launches; outs: tensors it writes"""
    global total_bad
    for o in outs:
        o.zero_()
    fn()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    bad, worst = 0, 0
    for _ in range(args.iters):
        for o in outs:
            o.fill_(7)              # (a stale value would show as well)
        fn()
        torch.cuda.synchronize()
        nds = [int((o.view(torch.uint8) != r.view(torch.uint8)).sum()) for o, r in zip(outs, ref)]
        nd = sum(nds)
        if nd:
            bad += 1
            worst = max(worst, nd)
            if bad <= 3 and detail is None and len(outs) > 1:
                print(f"    differing bytes per output: {nds}", flush=True)
                for o, r in zip(outs, ref):
                    if o.dim() == 3 and not torch.equal(o, r):        # (B, N, D) activations: which tokens?
                        tok = (o.view(torch.int16) != r.view(torch.int16)).any(dim=2).nonzero()
                        tl = sorted({(int(b_), int(n_) // 16) for b_, n_ in tok.tolist()})
                        print(f"      tokens differing: {tok.shape[0]}; (sample, 16-token tile): {tl[:24]}{' ...' if len(tl) > 24 else ''}", flush=True)
            if bad <= 2 and detail is not None:
                detail(outs, ref)
    total_bad += bad
    print(f"{name:48s} {bad:4d} of {args.iters} repetitions differ (worst: {worst} bytes)", flush=True)


ap_only = os.environ.get("CONTENTION_ATTN_ONLY") == "1"
for s, N in (() if ap_only else ((32, 49), (16, 196), (8, 784), (4, 3136))):       # (s = 4: Lite's finest level; CONTENTION_ATTN_ONLY=1 skips these)
    D = 3 * s * s
    npatch = B * N
    x = torch.randn(B, N, D, generator=g).to(bf).to(dev)
    w = [(torch.randn(3, 3, 3, 3, generator=g) / 5).to(dev) for _ in range(3)]
    q, k, v = (torch.empty_like(x) for _ in range(3))
    if not only or "conv" in only:
      repeat(f"conv3x3_qkv_fwd s={s} npatch={npatch}",
           lambda: check(L.vu_conv3x3_qkv_fwd(1, ptr(x), ptr(x), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(q), ptr(k), ptr(v), npatch, 3, s, st)), [q, k, v], detail=conv_detail(s))
    dq, dk, dv = (torch.randn(B, N, D, generator=g).to(bf).to(dev) for _ in range(3))
    dx = torch.empty_like(x)
    if not only or "dgrad" in only:
      repeat(f"conv3x3_qkv_dgrad s={s}",
           lambda: check(L.vu_conv3x3_qkv_dgrad(1, ptr(dq), ptr(dk), ptr(dv), ptr(w[0]), ptr(w[1]), ptr(w[2]), None, None, ptr(dx), None, npatch, 3, s, st)), [dx])
    if (not only or "wgrad" in only) and s % 8 == 0:      # the three weight gradients, partial sums through a lent slab (fixed order)
        dws = [torch.zeros(3, 3, 3, 3, device=dev) for _ in range(3)]
        scr = torch.empty(4 << 20, dtype=torch.uint8, device=dev)

        def wg():
            for t_ in dws:
                t_.zero_()
            check(L.vu_conv3x3_qkv_wgrad(1, ptr(dq), ptr(dk), ptr(dv), ptr(x), ptr(x), ptr(dws[0]), ptr(dws[1]), ptr(dws[2]), ptr(scr), scr.numel(), npatch, 3, s, st))
        repeat(f"conv3x3_qkv_wgrad s={s}", wg, dws)
    # residual add + LayerNorm over (N, D)
    P = N * D
    a = torch.randn(B, N, D, generator=g).to(bf).to(dev)
    lw, lb = torch.randn(P, generator=g).to(dev), torch.randn(P, generator=g).to(dev)
    z, y = torch.empty_like(a), torch.empty_like(a)
    lws = torch.empty(L.vu_layernorm_workspace_floats(B, P), dtype=torch.float32, device=dev)
    stats = torch.empty(2 * B, dtype=torch.float32, device=dev)
    if not only or "ln" in only:
      repeat(f"add_layernorm_fwd P={P}", lambda: check(L.vu_add_layernorm_fwd(1, ptr(a), ptr(x), ptr(z), ptr(lw), ptr(lb), ptr(y), ptr(lws), ptr(stats), B, P, st)), [z, y, stats])
    # re-tiling to the next level
    if s > 8 and (not only or "retile" in only):
        o = torch.empty_like(x)
        repeat(f"retile s={s}->{s // 2}", lambda: check(L.vu_retile(1, 0, 0, ptr(x), ptr(o), None, B, 3, 224, s, s // 2, st)), [o])
    # the level's projection: y = x W^T + b
    M = B * N
    wt = (torch.randn(D, D, generator=g) / D ** 0.5).to(bf).to(dev)
    yo = torch.empty(M, D, dtype=bf, device=dev)
    if not only or "gemm" in only:
      repeat(f"gemm M={M} N={D} K={D}", lambda: check(L.vu_gemm(1, 0, ptr(x), ptr(wt), ptr(yo), M, D, D, D, 1, 1, D, D, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, st)), [yo])
# the recompute attention of the Lite levels (4 heads; d = 12 padded to 16, d = 48) and of Base level 2 (8 heads, d = 24): forward +
# backward through the stand-alone op (its conv / projection WEIGHT gradients end in float atomics: compared are y, dx and the
# head-mix / BatchNorm gradients, which do not)
import ctypes as C
L.vu_set_attn_form(1, 0)
attn_cases = {"none": (), "h8": ((784, 3, 8, 8),), "all": ((3136, 3, 4, 4), (784, 3, 8, 4), (784, 3, 8, 8))}[args.attn]
for (N, Cn, s, H) in attn_cases:
    D = Cn * s * s
    Ba = max(2, B // 4)
    names = ["mix_w", "mix_b", "bn_w", "bn_b", "wq", "wk", "wv", "proj_w", "proj_b"]
    pd = {"mix_w": torch.eye(H) + 0.3 * torch.randn(H, H, generator=g), "mix_b": 0.05 * torch.randn(H, generator=g),
          "bn_w": 1 + 0.2 * torch.randn(H, generator=g), "bn_b": 0.1 * torch.randn(H, generator=g),
          "wq": torch.randn(Cn, Cn, 3, 3, generator=g) / (9 * Cn) ** 0.5, "wk": torch.randn(Cn, Cn, 3, 3, generator=g) / (9 * Cn) ** 0.5,
          "wv": torch.randn(Cn, Cn, 3, 3, generator=g) / (9 * Cn) ** 0.5, "proj_w": torch.randn(D, D, generator=g) / D ** 0.5,
          "proj_b": 0.05 * torch.randn(D, generator=g)}
    dd = {k_: v_.to(dev).contiguous() for k_, v_ in pd.items()}
    pw = dd["proj_w"].to(bf).contiguous()
    rm, rv = torch.zeros(H, device=dev), torch.ones(H, device=dev)
    prm = _lib.vu_attn_params(*[dd[k_].data_ptr() for k_ in names[:7]], pw.data_ptr(), dd["proj_b"].data_ptr(), rm.data_ptr(), rv.data_ptr())
    grads = [torch.zeros_like(dd[k_]) for k_ in names]
    gs = _lib.vu_attn_grads(*[t_.data_ptr() for t_ in grads])
    xa = torch.randn(Ba, N, D, generator=g).to(bf).to(dev)
    dya = torch.randn(Ba, N, D, generator=g).to(bf).to(dev)
    nbytes = L.vu_attn_workspace_bytes(1, Ba, N, D, H)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    ya, dxa = torch.empty_like(xa), torch.empty_like(xa)

    def attn_run():
        for t_ in grads[:4]:
            t_.zero_()
        rm.zero_(); rv.fill_(1.0)
        check(L.vu_attn_forward(1, C.byref(prm), ptr(xa), ptr(xa), ptr(ya), None, ptr(ws), nbytes, Ba, N, D, H, Cn, 0.2, 0.2, 1, 7, 3, st))
        check(L.vu_attn_backward(1, C.byref(prm), C.byref(gs), ptr(xa), ptr(xa), ptr(dya), ptr(dxa), None, ptr(ws), nbytes, Ba, N, D, H,
                                 Cn, 0.2, 0.2, 1, 7, 3, st))
    def attn_fwd():
        rm.zero_(); rv.fill_(1.0)
        check(L.vu_attn_forward(1, C.byref(prm), ptr(xa), ptr(xa), ptr(ya), None, ptr(ws), nbytes, Ba, N, D, H, Cn, 0.2, 0.2, 1, 7, 3, st))
    repeat(f"attention forward (recompute form) N={N} H={H} d={D // H} B={Ba}", attn_fwd, [ya])
    repeat(f"attention fwd+bwd (recompute form) N={N} H={H} d={D // H} B={Ba}", attn_run, [ya, dxa] + grads[:4])
L.vu_set_attn_form(-1, 0)
if child is not None:
    child.wait()
if load_thread is not None:
    load_thread.join(args.load + 60)
print("CONTENTION_OPS", "CLEAN" if total_bad == 0 else f"{total_bad} bad repetitions", flush=True)
sys.exit(0 if total_bad == 0 else 1)
