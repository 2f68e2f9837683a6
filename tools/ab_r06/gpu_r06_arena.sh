cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
for i in 1 2 3; do for B in 64 16 32; do
run 300 gpurun_out/r06ar_bench.log python bench.py --batch $B --no-cpu-baseline --no-host-input --no-sustained --dump-profile gpurun_out/r06ar_prof$B.json; echo "B=$B $(tail -1 gpurun_out/r06ar_bench.log | cut -c60-200)"
done; done
python - <<'PY'
import json
for B in (64,16,32):
    d=json.load(open(f"gpurun_out/r06ar_prof{B}.json"))
    print(B, {k[:30]:(v["count"], round(v["ms"]*1e3/v["count"],1)) for k,v in d.items() if "reduce_batch" in k})
PY
