cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06g_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3"; tail -3 gpurun_out/r06g_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rocprof_g -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py --B 64 --reps 40 > $GRAFT_REPO_ROOT/gpurun_out/r06g_cb.log 2>&1
cd $GRAFT_REPO_ROOT
grep "C=3" gpurun_out/r06g_cb.log | cut -c60-260
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/rocprof_g/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Name"]: print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/rocprof_g
