cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 600 gpurun_out/r06u_tests.log python -m pytest tests/test_gpu_ops.py tests/test_a_hotpath_gpu.py -x -q -k "flash or hotpath"; tail -3 gpurun_out/r06u_tests.log
for i in 1 2; do for V in 1 0; do
VU_FLASH_AP3=$V run 300 gpurun_out/r06u_bench$V.log python bench.py --no-cpu-baseline --no-host-input --no-sustained --dump-profile gpurun_out/r06u_prof$V.json; echo "AP3=$V $(tail -1 gpurun_out/r06u_bench$V.log | cut -c60-200)"
done; done
python - <<'PY'
import json
for f in ("1","0"):
    d=json.load(open(f"gpurun_out/r06u_prof{f}.json"))
    print(f, {k[:24]:round(v["ms"]*1e3/v["count"],1) for k,v in d.items() if "apply" in k or "moments" in k or "rowstats" in k})
PY
