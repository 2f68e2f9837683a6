cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06e_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3"; tail -12 gpurun_out/r06e_tests.log
run 300 gpurun_out/r06e_hot.log python -m pytest tests/test_a_hotpath_gpu.py -x -q; tail -3 gpurun_out/r06e_hot.log
run 300 gpurun_out/r06e_bench.log python bench.py --no-cpu-baseline --no-host-input --no-sustained --dump-profile gpurun_out/r06e_prof_new.json; tail -1 gpurun_out/r06e_bench.log | cut -c1-330
VU_CONV_TZ=0 run 300 gpurun_out/r06e_bench_old.log python bench.py --no-cpu-baseline --no-host-input --no-sustained --dump-profile gpurun_out/r06e_prof_old.json; tail -1 gpurun_out/r06e_bench_old.log | cut -c1-330
python - <<'PY'
import json
for f in ("new","old"):
    d=json.load(open(f"gpurun_out/r06e_prof_{f}.json"))
    print(f, {k:(round(v["ms_per_step"]*1e3/ (v["count"]/2),1), v["count"]//2) for k,v in d.items() if "conv" in k})
PY
