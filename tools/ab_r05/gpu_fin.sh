cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_a_hotpath_gpu.py -x -q -m gpu -k "attention or flash or conv or poison" 2>&1 | tail -2
for b in 16 64; do
  timeout -k 10 200 python tools/step_tags.py --batch $b --grep "inalize" 2>&1 | grep -E "inalize|sum of" 
  timeout -k 10 200 python tools/step_tags.py --batch $b --grep "reduce" 2>&1 | grep -E "reduce"
done
for b in 64 16; do timeout -k 10 300 python bench.py --batch $b --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/f.log 2>&1 && tail -1 gpurun_out/f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH B=$b', round(d['value'],1), round(d['ms_per_step'],4))"; done
timeout -k 10 300 python bench.py --model lite --batch 32 --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/f.log 2>&1 && tail -1 gpurun_out/f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH lite', round(d['value'],1), round(d['ms_per_step'],4))"
