"""Copies the summaries tools/gpu_round4.sh left in gpurun_out/ into profiles/ (tracked): bench lines, rocprofv3 kernel stats,
per-grid table, the FETCH / WRITE PMC passes joined into r04_pmc_traffic.json, MFMA busy, SQ counters of the recompute sweeps."""
import glob
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
for f in glob.glob(os.path.join(G, "r04_bench_*.json")):
    shutil.copy(f, P)
for src, dst in [("r04_kernel_stats.csv", "r04_kernel_stats.csv"), ("r04_by_grid.csv", "r04_by_grid.csv"),
                 ("pmc_mfma_summary.csv", "r04_pmc_mfma_util_summary.csv"), ("r04_gaps.txt", "r04_gaps.txt"), ("r04_flash_pmc_sq_summary.csv", "r04_flash_pmc_sq_summary.csv")]:
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    else:
        print("missing", src)
if os.path.exists(os.path.join(G, "pmc_FETCH_SIZE_summary.csv")):
    subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic_json.py"), "r04"], cwd=R)
