#!/bin/bash
# Register / spill / occupancy report of the recompute-attention kernels (hipcc -Rpass-analysis=kernel-resource-usage), no GPU needed.
#   bash tools/flash_regs.sh [grep pattern, default flash2]
cd "$(dirname "$0")/../vit-unet_amd/csrc"
PAT=${1:-flash2}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-honor-nans -Rpass-analysis=kernel-resource-usage --cuda-device-only -c vu_flash.hip -o /tmp/vu_flash_regs.o 2>&1 |
python3 -c "
import sys, re, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': subprocess.run(['/usr/bin/c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur); continue
    for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('spill', r'VGPRs Spill: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
for r in rows:
    n = r['name']
    if '$PAT' not in n: continue
    n = re.sub(r'^void \(anonymous namespace\)::', '', n); n = re.sub(r'\(.*$', '', n)
    print(f\"{n:64s} vgpr {r.get('vgpr', -1):4d} agpr {r.get('agpr', -1):4d} spill {r.get('spill', -1):4d} scratch {r.get('scratch', -1):5d} occ {r.get('occ', -1)}\")
"
