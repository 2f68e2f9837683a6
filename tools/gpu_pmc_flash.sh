# SQ counter passes over tools/flash_bench.py (the stand-alone attention op, flash form): where the waves spend their
# cycles.  Usage on the GPU box:  bash tools/gpu_pmc_flash.sh [tag]
TAG=${1:-flash}
cd /tmp && export TMPDIR=/tmp
run_pass() {   # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -- python3 $GRAFT_REPO_ROOT/tools/flash_bench.py --reps 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log 2>&1
  tail -c 300 $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log
}
run_pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT
run_pass b SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
run_pass c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16
cd $GRAFT_REPO_ROOT
python - "$TAG" <<'PY'
import csv, glob, collections, sys, shutil
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for p in "abc":
    for f in glob.glob(f"gpurun_out/pmc_{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            import re
            m = re.search(r"(flash\d?_\w+?)(<[^>]*>)?\(", r["Kernel_Name"] + "(")
            if not m: continue
            k = m.group(1) + (m.group(2) or "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
names = sorted({n for v in agg.values() for n in v})
with open(f"gpurun_out/{tag}_pmc_sq_summary.csv", "w") as o:
    o.write("kernel," + ",".join(names) + "\n")
    for k, v in sorted(agg.items()):
        o.write('"%s",' % k + ",".join("%.0f" % (v[n] / max(cnt[k][n], 1)) if n in v else "" for n in names) + "\n")
print(open(f"gpurun_out/{tag}_pmc_sq_summary.csv").read())
for p in "abc": shutil.rmtree(f"gpurun_out/pmc_{p}", ignore_errors=True)
PY
