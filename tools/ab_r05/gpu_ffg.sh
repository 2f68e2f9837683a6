cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity_full.py -x -q -m gpu -k "feedforward" 2>&1 | tail -2
timeout -k 10 600 python -m pytest tests/test_a_hotpath_gpu.py tests/test_gpu_parity_full.py -x -q -m gpu -k "lite" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do
  VU_FF2=$v timeout -k 10 300 python bench.py --model lite --batch 32 --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/f.log 2>&1 && tail -1 gpurun_out/f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH lite FF2=$v', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
timeout -k 10 200 python tools/step_tags.py --model lite --batch 32 --grep "ff2" 2>&1 | grep ff2
