cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06n_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "centred_map_form"; tail -12 gpurun_out/r06n_tests.log | cut -c1-200
run 300 gpurun_out/r06n_bench.log python bench.py --no-cpu-baseline --no-host-input --no-sustained --dump-profile gpurun_out/r06n_prof.json; tail -1 gpurun_out/r06n_bench.log | cut -c60-230
VU_ATTN_F1=0 run 300 gpurun_out/r06n_bench0.log python bench.py --no-cpu-baseline --no-host-input --no-sustained --dump-profile gpurun_out/r06n_prof0.json; tail -1 gpurun_out/r06n_bench0.log | cut -c60-230
python - <<'PY'
import json
for f in ("","0"):
    d=json.load(open(f"gpurun_out/r06n_prof{f}.json"))
    print(f or "f1", {k[:28]:(round(v["ms"]*1e3/v["count"],1), v["count"]//2) for k,v in d.items() if any(t in k for t in ("attn_f1","scores","mix_","bn_fin","map_rows"))})
PY
