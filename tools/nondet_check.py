"""Bisect tool for run-to-run differences of the Base bf16 train step (round 4).

    python tools/nondet_check.py [--B 20] [--reps 4] [--poison] [--fresh] [--load SECONDS] [--model base]

Every repetition runs forward + loss + backward of the SAME step (same weights, batch, dropout seed, step counter) through
the fused engine and compares the output image and every parameter gradient with repetition 0, bit for bit.
  --poison : before each repetition the model workspace is filled with a different byte pattern (0x00, 0xFF = NaN, 0x7F,
             random): an uninitialised read shows up as a difference (or as NaN).
  --fresh  : every repetition builds a new model + engine (what tools/dp_rehearsal.py --base does for its reference sums).
  --load S : a second process runs the same kind of step on the same GPU for S seconds while the repetitions run (the only
             structural difference between the two legs of the two-rank rehearsal that went red on the driver's box).
Environment switches of the library (VU_BGEMM, VU_CONV_W, VU_ATTN_FLASH, ...) are inherited, so the caller bisects with them."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=20)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--poison", action="store_true")
ap.add_argument("--fresh", action="store_true")
ap.add_argument("--load", type=float, default=0.0)
ap.add_argument("--model", default="base")
ap.add_argument("--alloc-poison", type=float, default=0.0, help="GiB of poisoned memory handed back to the caching allocator before each repetition")
ap.add_argument("--twice", action="store_true")
ap.add_argument("--ws-diff", type=int, default=0, help="N: run the FORWARD N + 1 times on one model and name the workspace buffers that differ from run 0")
ap.add_argument("--as-load", type=float, default=0.0, help=argparse.SUPPRESS)
args = ap.parse_args()

child = None
if args.load > 0:      # started before this process touches the GPU
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--as-load", str(args.load), "--B", str(args.B),
                              "--model", args.model])

sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd"))
import torch  # noqa: E402
from vit_unet.torch import model as M  # noqa: E402
from vit_unet.torch.engine import TrainStep  # noqa: E402


def build():
    torch.manual_seed(0)
    m = M.get_vit_unet(args.model, dtype=torch.bfloat16).to("cuda").train()
    return m, TrainStep(m, lr=1e-4, seed=7)


def batch():
    g = torch.Generator().manual_seed(1000)
    c, im = (1, 512) if args.model == "seg512" else (3, 224)
    y = torch.rand(args.B, c, im, im, generator=g)
    return (y + 0.1 * torch.randn(y.shape, generator=g)).clamp(0, 1).cuda(), y.cuda()


if args.as_load > 0:
    m, ts = build()
    x, y = batch()
    t0 = time.time()
    n = 0
    while time.time() - t0 < args.as_load:
        ts.step(x, y)
        torch.cuda.synchronize()
        n += 1
        if n == 1:
            open(f"/tmp/nondet_load_{os.getpid()}", "w").close()
    print(f"[load] {n} steps", flush=True)
    sys.exit(0)

x, y = batch()
m, ts = build()
if child is not None:          # do not start before the load process really runs steps on the GPU
    import glob
    t0 = time.time()
    while not glob.glob(f"/tmp/nondet_load_{child.pid}") and time.time() - t0 < 120:
        time.sleep(0.2)
    print(f"load process running after {time.time() - t0:.1f} s", flush=True)
if args.ws_diff:
    import ctypes as C
    from vit_unet.torch._lib import lib
    buf = C.create_string_buffer(1 << 17)
    lib().vu_model_workspace_describe(C.byref(m._cfg), args.B, buf, len(buf))
    layout = [(n_, int(o), int(b)) for n_, o, b in (ln.split() for ln in buf.value.decode().splitlines())]
    ws = m._workspace(args.B)
    ref_ws = None
    bad_runs = 0
    for rep in range(args.ws_diff + 1):
        ts.step_count.zero_()
        ws.zero_()
        out, dout = torch.empty_like(x), torch.empty_like(x)
        ts._enqueue_head(x, y, out, dout)
        torch.cuda.synchronize()
        if ref_ws is None:
            ref_ws = ws.clone()
            continue
        if torch.equal(ws, ref_ws):
            continue
        bad_runs += 1
        shown = 0
        for n_, o, b in layout:
            if b == 0:
                continue
            d = ws[o:o + b] != ref_ws[o:o + b]
            nd = int(d.sum())
            if nd:
                first = int(d.nonzero()[0])
                print(f"  run {rep}: {n_} differs in {nd} of {b} bytes (first at byte {first})", flush=True)
                shown += 1
                if shown >= 6:
                    break
    if child is not None:
        child.wait()
    print("NONDET_CHECK", "IDENTICAL" if bad_runs == 0 else f"DIFFERS in {bad_runs} of {args.ws_diff} runs", flush=True)
    sys.exit(0 if bad_runs == 0 else 1)
patterns = [0x00, 0xFF, 0x7F, None]
ref = None
names = None
worst = 0
for rep in range(args.reps):
    if args.fresh and rep > 0:
        del m, ts
        m, ts = build()
    ts.step_count.zero_()
    ws = m._workspace(args.B)
    if args.poison:
        p = patterns[rep % 4]
        if p is None:
            ws.copy_(torch.randint(0, 256, (ws.numel(),), dtype=torch.uint8, device="cuda"))
        else:
            ws.fill_(p)
    if args.alloc_poison:     # everything the caching allocator hands out next is poisoned too (not only the workspace)
        junk = torch.empty(int(args.alloc_poison * 2 ** 30), dtype=torch.uint8, device="cuda")
        junk.fill_(0x7F if rep % 2 else 0xCB)
        del junk
    out, dout = torch.empty_like(x), torch.empty_like(x)
    sums = (m._arena.double().sum().item(), m._shadow.float().double().sum().item(), m._bn.double().sum().item(),
            x.double().sum().item(), y.double().sum().item())
    ts._enqueue_head(x, y, out, dout)
    ts._enqueue_units(dout, 0, ts._nunits - 1)
    torch.cuda.synchronize()
    g = m._garena.detach().clone()
    if args.twice:            # the same step once more on the same objects: a wrong result that repeats is state, one that does not is a race
        ts.step_count.zero_()
        out2, dout2 = torch.empty_like(x), torch.empty_like(x)
        ts._enqueue_head(x, y, out2, dout2)
        ts._enqueue_units(dout2, 0, ts._nunits - 1)
        torch.cuda.synchronize()
        print(f"   rep {rep} run twice: outputs equal {torch.equal(out, out2)}, gradients equal {torch.equal(g, m._garena)}; "
              f"checksums arena {sums[0]:.6f} shadow {sums[1]:.6f} bn {sums[2]:.4f} x {sums[3]:.4f} y {sums[4]:.4f}", flush=True)
    if ref is None:
        ref = (out.clone(), g)
        base_ptr = m._arena.data_ptr()
        names = [(n_, (p_.data_ptr() - base_ptr) // 4, p_.numel()) for n_, p_ in m.named_parameters()]
        print(f"rep 0: loss {ts.loss.item():.6f}  finite out {bool(torch.isfinite(out).all())}  finite grads {bool(torch.isfinite(g).all())}",
              flush=True)
        continue
    nout = int((out != ref[0]).sum())
    ng = int((g != ref[1]).sum())
    worst = max(worst, nout + ng)
    print(f"rep {rep}: output elements differing {nout} of {out.numel()} (max |diff| {(out - ref[0]).abs().max().item():.3e}); "
          f"gradient elements differing {ng} of {g.numel()}  finite {bool(torch.isfinite(g).all())}", flush=True)
    if ng:
        shown = 0
        for n_, o, k in names:
            d = g[o:o + k] != ref[1][o:o + k]
            if d.any() and shown < 12:
                e = (g[o:o + k] - ref[1][o:o + k]).abs().max().item()
                print(f"     {n_}: {int(d.sum())} of {k}, max |diff| {e:.3e} of max |g| {ref[1][o:o + k].abs().max().item():.3e}", flush=True)
                shown += 1
if child is not None:
    child.wait()
print("NONDET_CHECK", "IDENTICAL" if worst == 0 else "DIFFERS", flush=True)
sys.exit(0 if worst == 0 else 1)
