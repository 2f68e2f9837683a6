"""Copies the summaries tools/gpu_round6.sh left in gpurun_out/ into profiles/ (tracked): bench lines, rocprofv3 kernel stats,
per-grid table, the FETCH / WRITE PMC passes joined into r06_pmc_traffic.json, MFMA busy, SQ counters of the recompute sweeps."""
import glob
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
for f in glob.glob(os.path.join(G, "r06_bench_*.json")):
    shutil.copy(f, P)
for src, dst in [("r06_kernel_stats.csv", "r06_kernel_stats.csv"), ("r06_by_grid.csv", "r06_by_grid.csv"),
                 ("pmc_mfma_summary.csv", "r06_pmc_mfma_util_summary.csv"), ("r06_gaps.txt", "r06_gaps.txt"), ("r06_flash_pmc_sq_summary.csv", "r06_flash_pmc_sq_summary.csv")]:
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    else:
        print("missing", src)
if os.path.exists(os.path.join(G, "r06_base_64_pmc_FETCH_SIZE_summary.csv")):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(G, f"r06_base_64_pmc_{c}_summary.csv"), os.path.join(G, f"pmc_{c}_summary.csv"))
    subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic_json.py"), "r06"], cwd=R)
# the flash-family PMC passes of the other BASELINE configurations (tools/gpu_round6.sh pmc2)
for model, batch in (("lite", 32), ("base", 16), ("large", 16)):
    ok = True
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        src = os.path.join(G, f"r06_{model}_{batch}_pmc_{c}_summary.csv")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(G, f"pmc_{c}_summary.csv"))
        else:
            ok = False
    if ok:
        subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic_json.py"), f"r06b{batch}_{model}", model, str(batch)], cwd=R)
