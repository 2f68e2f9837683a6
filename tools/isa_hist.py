"""Instruction histogram of the hot loop(s) of a kernel: compiles one .hip file of csrc/ to gfx950 assembly (hipcc -S, no GPU
needed) and prints, for every basic block of the chosen kernel holding at least --min-mfma matrix instructions, the count
per opcode.  This is how the round-2 VALU diet of the recompute attention sweeps was found (canonicalising v_max, selects
for dead k-slots, quarter-rate v_mad_u64_u32 in the mask hash, ...).
usage: python tools/isa_hist.py vu_flash.hip flash2_bwd_dqx_kernelILi24 [--flags=-fno-honor-nans] [--min-mfma 30]"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument("file")
ap.add_argument("kernel", help="substring of the mangled kernel name")
ap.add_argument("--flags", default="")
ap.add_argument("--min-mfma", type=int, default=8)
ap.add_argument("--top", type=int, default=30)
a = ap.parse_args()
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-unet_amd", "csrc")
out = os.path.join(tempfile.gettempdir(), os.path.basename(a.file) + ".s")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only"] + a.flags.split() + [os.path.join(root, a.file), "-o", out]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
starts = [i for i, l in enumerate(lines) if a.kernel in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S+:", l) and a.kernel in l.split(":")[0]]
if not starts:
    sys.exit("kernel not found")
for st in starts:
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    print(lines[st].split(":")[0])
    blocks, cur, name = [], [], "entry"
    for l in lines[st + 1:end]:
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append((name, cur)); name, cur = t.split(":")[0], []
        elif t and not t.startswith((";", ".")):
            cur.append(t.split()[0])
    blocks.append((name, cur))
    for n, b in blocks:
        c = collections.Counter(b)
        mf = sum(v for k, v in c.items() if "mfma" in k)
        if mf < a.min_mfma:
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
        print(f"  {n}: {len(b)} instructions, VALU {valu}, MFMA {mf}, DS {sum(v for k, v in c.items() if k.startswith('ds_'))}, "
              f"SALU {sum(v for k, v in c.items() if k.startswith('s_'))}, VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_')))}")
        print("     " + ", ".join(f"{k} {v}" for k, v in c.most_common(a.top)))
