set -x
cd $GRAFT_REPO_ROOT
for cfg in "base 16" "large 16" "lite 32" "base 32"; do set -- $cfg
for mw in 32 30 28; do
  VU_BGEMM_MINWORK=$mw timeout -k 10 300 python bench.py --model $1 --batch $2 --steps 30 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/mw.log 2>&1 && tail -1 gpurun_out/mw.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MINWORK=$mw $1 $2', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
