"""Copies the summaries tools/gpu_round5.sh left in gpurun_out/ into profiles/ (tracked): bench lines, rocprofv3 kernel stats,
per-grid table, the FETCH / WRITE PMC passes joined into r05_pmc_traffic.json, MFMA busy, SQ counters of the recompute sweeps."""
import glob
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
for f in glob.glob(os.path.join(G, "r05_bench_*.json")):
    shutil.copy(f, P)
for src, dst in [("r05_kernel_stats.csv", "r05_kernel_stats.csv"), ("r05_by_grid.csv", "r05_by_grid.csv"),
                 ("pmc_mfma_summary.csv", "r05_pmc_mfma_util_summary.csv"), ("r05_gaps.txt", "r05_gaps.txt"), ("r05_flash_pmc_sq_summary.csv", "r05_flash_pmc_sq_summary.csv")]:
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    else:
        print("missing", src)
if os.path.exists(os.path.join(G, "r05_base_64_pmc_FETCH_SIZE_summary.csv")):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(G, f"r05_base_64_pmc_{c}_summary.csv"), os.path.join(G, f"pmc_{c}_summary.csv"))
    subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic_json.py"), "r05"], cwd=R)
# the flash-family PMC passes of the other BASELINE configurations (tools/gpu_round5.sh pmc2)
for model, batch in (("lite", 32), ("base", 16), ("large", 16)):
    ok = True
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        src = os.path.join(G, f"r05_{model}_{batch}_pmc_{c}_summary.csv")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(G, f"pmc_{c}_summary.csv"))
        else:
            ok = False
    if ok:
        subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic_json.py"), f"r05b{batch}_{model}", model, str(batch)], cwd=R)
