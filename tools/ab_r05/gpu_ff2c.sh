cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity_full.py -x -q -m gpu -k "feedforward" 2>&1 | tail -1
for b in 64 16; do timeout -k 10 120 python tools/ff2_time.py $b 2>&1 | grep -E "ff2_bwd"; timeout -k 10 200 python tools/step_tags.py --batch $b --grep "ff2" 2>&1 | grep ff2; done
timeout -k 10 400 python -m pytest tests/test_a_hotpath_gpu.py tests/test_gpu_parity_full.py -x -q -m gpu -k "teacher_forced or poison" 2>&1 | tail -1
for b in 64 16; do timeout -k 10 300 python bench.py --batch $b --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/f.log 2>&1 && tail -1 gpurun_out/f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH B=$b', round(d['value'],1), round(d['ms_per_step'],4))"; done
