cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
VU_MAP_BWD_MM_MINLD=56 run 400 gpurun_out/r06p_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "(test_attention_fwd_bwd or centred_map_form) and 49-3-32-8"; tail -4 gpurun_out/r06p_tests.log | cut -c1-200
for V in 64 56; do
VU_MAP_BWD_MM_MINLD=$V run 300 gpurun_out/r06p_bench$V.log python bench.py --no-cpu-baseline --no-host-input --no-sustained --dump-profile gpurun_out/r06p_prof$V.json; tail -1 gpurun_out/r06p_bench$V.log | cut -c60-230
done
python - <<'PY'
import json
for f in ("64","56"):
    d=json.load(open(f"gpurun_out/r06p_prof{f}.json"))
    print(f, {k[:28]:(round(v["ms"]*1e3/v["count"],1), v["count"]//2) for k,v in d.items() if any(t in k for t in ("map_bwd","attn_f1"))})
PY
