set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_a_hotpath_gpu.py -x -q -m gpu -k "attention or poison or flash" > gpurun_out/lb_tests.txt 2>&1 || { tail -30 gpurun_out/lb_tests.txt; exit 1; }
tail -2 gpurun_out/lb_tests.txt
for b in 64 16; do for v in 0 1; do
  VU_LAST_BLOCK=$v timeout -k 10 200 python tools/step_tags.py --batch $b --grep "${LB_GREP:-bn_bwd_small}" 2>&1 | grep -E "LAST_BLOCK|sum of tags"
done; done
for rep in 1 2; do for v in 0 1; do for b in 64 16; do
  VU_LAST_BLOCK=$v timeout -k 10 300 python bench.py --batch $b --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/lb_bench.log 2>&1 && tail -1 gpurun_out/lb_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH LAST_BLOCK=$v B=$b', round(d['value'],1), round(d['ms_per_step'],4))"
done; done; done
