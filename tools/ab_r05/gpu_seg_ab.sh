cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "seg512 8" "seg512 32" "base 128"; do set -- $cfg
  for v in old new; do
    if [ $v = old ]; then export VU_BGEMM_MINWORK=32 VU_BGEMM_SHORTK_M=2048 VU_BGEMM_SHORTK_MN=22 VU_GEMM_3232_BK=64 VU_GEMM_3264_BK=64; else unset VU_BGEMM_MINWORK VU_BGEMM_SHORTK_M VU_BGEMM_SHORTK_MN VU_GEMM_3232_BK VU_GEMM_3264_BK; fi
    timeout -k 10 300 python bench.py --model $1 --batch $2 --steps 30 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/ab.log 2>&1 && tail -1 gpurun_out/ab.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB $v $1 $2', round(d['value'],1), round(d['ms_per_step'],4))"
  done; done; done
