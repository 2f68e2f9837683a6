set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -2
for b in 16 64; do for v in 0 1; do
  VU_BN_BWD_WIDE=$v timeout -k 10 200 python tools/step_tags.py --batch $b --grep "bn_bwd_small" 2>&1 | grep -E "bn_bwd" | sed "s/^/WIDE=$v /"
done; done
for rep in 1 2; do for v in 0 1; do
  VU_BN_BWD_WIDE=$v timeout -k 10 300 python bench.py --batch 16 --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/bw_bench.log 2>&1 && tail -1 gpurun_out/bw_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH WIDE=$v B=16', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
