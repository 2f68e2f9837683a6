"""profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries tools/gpu_pmc.sh leaves in
gpurun_out/ (usage: python tools/pmc_traffic_json.py r05 [model batch]; for another workload use a tag that ends in the model
name, e.g. r05_lite: bench.py looks for profiles/*_<model>_pmc_traffic.json first and checks the workload field)."""
import csv
import json
import shutil
import sys

tag = sys.argv[1]
model = sys.argv[2] if len(sys.argv) > 2 else "base"         # the workload the PMC passes ran (BENCH_ARGS of tools/gpu_pmc.sh)
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
k = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy(f"gpurun_out/pmc_{c}_summary.csv", f"profiles/{tag}_pmc_{c}_summary.csv")
    for r in csv.DictReader(open(f"gpurun_out/pmc_{c}_summary.csv")):
        k.setdefault(r["kernel"], {})[c] = float(r["avg_per_launch"])
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_pmc.sh), KB per launch averaged over "
               "all launches of the symbol (a symbol that runs at several token levels averages over them); "
               "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts half the bytes of wide "
               "coalesced reads (MI355X_MICROARCH.md, HBM)",
       "workload": {"model": model, "batch": batch},
       "kernels": {}}
missing = []
for name, v in sorted(k.items()):
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    if f is None or w is None:        # (a symbol seen in one pass only: a launch-count difference between the two runs)
        missing.append(name)
        continue
    out["kernels"][name] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "traffic_bytes": (2 * f + w) * 1024}
out["symbols_in_one_pass_only"] = missing
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(len(out["kernels"]), "symbols joined,", len(missing), "in one pass only")
print(json.dumps({n[:60]: r["traffic_bytes"] for n, r in out["kernels"].items() if "flash2_bwd_dqx" in n}))
