"""Round 6, review item 6: WHOSE weights does conv_fwd_kernel's scalar-load form read when it goes wrong beside a second process?

    VU_CONV_W=smem python tools/sharing_alias.py [--secs 25] [--s 16] [--B 20]

Two child processes (the parent never touches the GPU) run the SAME op - vu_conv3x3_qkv_fwd with scalar-load weights, the failing
build kept from round 4 - on the same input, with DIFFERENT weights (seeds 1 and 2), allocated in the same order (so the weight
tensors very likely sit at the same VIRTUAL addresses in both processes; each child prints them).  Each child first computes, alone
on the GPU, its own reference and the output it WOULD get with its partner's weights; then both loop together and every element that
differs from the own reference is classified: equal to the partner-weights output (the scalar cache handed this wave the OTHER
process's data for the same virtual address) or something else (stale / torn data: a save / restore or ordering problem)."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--secs", type=float, default=25.0)
ap.add_argument("--s", type=int, default=16)
ap.add_argument("--B", type=int, default=20)
ap.add_argument("--child", type=int, default=0)
ap.add_argument("--tag", default="")
args = ap.parse_args()

if not args.child:
    tag = f"/tmp/sharing_alias_{os.getpid()}"
    kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(i), "--tag", tag, "--secs", str(args.secs),
                              "--s", str(args.s), "--B", str(args.B)]) for i in (1, 2)]
    rc = [k.wait() for k in kids]
    sys.exit(max(rc))

sys.path.insert(0, os.path.join(ROOT, "vit-unet_amd"))
import torch  # noqa: E402
from vit_unet.torch import _lib  # noqa: E402
from vit_unet.torch._lib import check, lib, ptr  # noqa: E402

L = lib()
me, other = args.child, 3 - args.child
s, B = args.s, args.B
N = (224 // s) ** 2
D = 3 * s * s
npatch = B * N
st = _lib.stream_ptr()
x = torch.randn(B, N, D, generator=torch.Generator().manual_seed(99)).to(torch.bfloat16).cuda()
q, k, v = (torch.empty_like(x) for _ in range(3))


def weights(seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(3, 3, 3, 3, generator=g) / 5) for _ in range(3)]


w_slot = [t.cuda() for t in weights(me)]          # the SAME three allocations in both processes; contents differ


def run():
    check(L.vu_conv3x3_qkv_fwd(1, ptr(x), ptr(x), ptr(w_slot[0]), ptr(w_slot[1]), ptr(w_slot[2]), ptr(q), ptr(k), ptr(v), npatch, 3, s, st))
    torch.cuda.synchronize()
    return torch.cat([q.view(torch.int16).reshape(-1), k.view(torch.int16).reshape(-1), v.view(torch.int16).reshape(-1)]).clone()


# phase 1, one child at a time (child 2 waits for child 1's file): references with the own and with the partner's weights
if me == 2:
    t0 = time.time()
    while not os.path.exists(args.tag + "_ref1") and time.time() - t0 < 120:
        time.sleep(0.1)
for t_, w_ in zip(w_slot, weights(other)):
    t_.copy_(w_.cuda())
ref_other = run()
for t_, w_ in zip(w_slot, weights(me)):
    t_.copy_(w_.cuda())
ref = run()
again = run()
print(f"[child {me}] weight tensors at {[hex(t.data_ptr()) for t in w_slot]}, x at {hex(x.data_ptr())}; alone: repeat identical = {bool(torch.equal(ref, again))}; "
      f"own vs partner-weights outputs differ in {int((ref != ref_other).sum())} of {ref.numel()} elements", flush=True)
open(args.tag + f"_ref{me}", "w").close()
t0 = time.time()
while not (os.path.exists(args.tag + "_ref1") and os.path.exists(args.tag + "_ref2")) and time.time() - t0 < 120:
    time.sleep(0.05)
# phase 2: both loop
t0 = time.time()
reps = bad = n_bad_el = n_partner = 0
while time.time() - t0 < args.secs:
    out = run()
    reps += 1
    d = out != ref
    nd = int(d.sum())
    if nd:
        bad += 1
        n_bad_el += nd
        n_partner += int((out[d] == ref_other[d]).sum())
print(f"[child {me}] s={s}: {bad} of {reps} repetitions differ; {n_bad_el} wrong elements, of which {n_partner} EQUAL the output computed with the "
      f"partner process's weights", flush=True)
