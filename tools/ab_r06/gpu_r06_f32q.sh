cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
for i in 1 2; do for V in "1 1024" "0 1024" "32 1024" "0 768"; do set -- $V
VU_GEMM_F32_QUARTER=$1 VU_TSGEMM_MINK=$2 run 300 gpurun_out/r06q_bench.log python bench.py --batch 16 --no-cpu-baseline --no-host-input --no-sustained --dump-profile gpurun_out/r06q_prof.json; echo "Q=$1 MINK=$2 $(tail -1 gpurun_out/r06q_bench.log | cut -c60-160)"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06q_prof.json"))
print("   ", {k[:34]:(v["count"], round(v["ms"]*1e3/v["count"],1)) for k,v in d.items() if ("f32" in k and "vu_gemm" in k) or "x128>" in k or "128x" in k})
PY
done; done
VU_GEMM_F32_QUARTER=1 run 300 gpurun_out/r06q_bench.log python bench.py --model large --batch 16 --no-cpu-baseline --no-host-input --no-sustained --no-roofline; echo "large Q=1 $(tail -1 gpurun_out/r06q_bench.log | cut -c60-160)"
VU_GEMM_F32_QUARTER=0 run 300 gpurun_out/r06q_bench.log python bench.py --model large --batch 16 --no-cpu-baseline --no-host-input --no-sustained --no-roofline; echo "large Q=0 $(tail -1 gpurun_out/r06q_bench.log | cut -c60-160)"
