"""Scans the gfx950 code objects of the library for the instruction pattern tools/probe/mfma_chain_probe.hip shows to be unsafe
as hipcc 7.2 schedules it: a matrix instruction whose accumulator input (SrcC) is the destination of a DIFFERENT matrix opcode
issued fewer than WINDOW instructions earlier (same-opcode chains forward in hardware and are fine).

    python tools/mfma_chain_scan.py [file.hip ...]          (default: every .hip of vit-unet_amd/csrc; a few minutes of hipcc -S)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = 12
rx = re.compile(r"^(v_mfma_\w+)\s+(\S+?),\s*(\S+?),\s*(\S+?),\s*(\S+?)(?:\s|$)")


def regs(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return tok[0], int(m.group(1)), int(m.group(2))
    m = re.match(r"([va])(\d+)$", tok)
    return (m.group(1), int(m.group(2)), int(m.group(2))) if m else None


def scan(path):
    """path: a .hip source of csrc/ - compiled to gfx950 assembly with the Makefile's flags (hipcc -S, no GPU needed)"""
    out = os.path.join(tempfile.gettempdir(), os.path.basename(path) + ".scan.s")
    flags = ["-fno-honor-nans"] if os.path.basename(path) == "vu_flash.hip" else []
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only"] + flags + [path, "-o", out],
                       capture_output=True, text=True)
    if r.returncode != 0:
        return None
    txt = open(out).read()
    os.remove(out)
    hits, fn, recent = [], "?", []          # recent: (opcode, dst regs, age in instructions)
    for line in txt.splitlines():
        m = re.match(r"^(_Z\S+):", line)
        if m:
            fn, recent = m.group(1), []
            continue
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        m = rx.match(t)
        nop = re.match(r"^s_nop\s+(\d+)", t)
        step = int(nop.group(1)) + 1 if nop else 1               # (s_nop n = n + 1 wait states)
        recent = [(o, d, age + step) for o, d, age in recent if age + step < WINDOW]
        if not m:
            continue
        op, dst, srcc = m.group(1), regs(m.group(2)), regs(m.group(5))
        if srcc:
            for pop, pd, age in recent:
                if pop != op and pd and pd[0] == srcc[0] and not (pd[2] < srcc[1] or srcc[2] < pd[1]):
                    hits.append((fn, pop, op, age))
        recent.append((op, dst, 0))
    return hits


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "vit-unet_amd", "csrc", "*.hip")))
    total = 0
    for f in files:
        h = scan(f)
        if h is None:
            print(f"{os.path.basename(f)}: did not compile")
            total += 1
            continue
        n = len(h)
        total += n
        print(f"{os.path.basename(f)}: {n} mixed-opcode accumulator chains within {WINDOW} instructions")
        seen = set()
        for fn, a, b, dist in h:
            key = (fn[:90], a, b)
            if key not in seen and len(seen) < 6:
                seen.add(key)
                print(f"    {fn[:90]}: {a} -> {b} ({dist} instructions apart)")
    print("MFMA_CHAIN_SCAN", "CLEAN" if total == 0 else f"{total} sites")
    return 0 if total == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
