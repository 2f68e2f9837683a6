# SQ counters for the dominant kernels (one --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-input --no-graph --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1
tail -c 200 $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:14]
with open("gpurun_out/pmc_sq_summary.csv", "w") as o:
    names = ["SQ_WAVES","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_INSTS_VALU","SQ_ACTIVE_INST_VALU","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_LDS_BANK_CONFLICT"]
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k, v in rows:
        o.write('"%s",%d,' % (k, cnt[k]) + ",".join("%.0f" % (v.get(n, 0) / max(cnt[k], 1)) for n in names) + "\n")
print(open("gpurun_out/pmc_sq_summary.csv").read())
import shutil; shutil.rmtree("gpurun_out/pmc_sq", ignore_errors=True)
PY
