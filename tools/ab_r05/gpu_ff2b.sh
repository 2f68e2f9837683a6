set -x
cd $GRAFT_REPO_ROOT
for v in 0 1; do for b in 64 16; do VU_FF2=$v timeout -k 10 120 python tools/ff2_time.py $b 2>&1 | grep -E "VU_FF2=" ; done; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity_full.py -x -q -m gpu -k "feedforward" 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
  VU_FF2=$v timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/ff2_bench_$v.log 2>&1 && tail -1 gpurun_out/ff2_bench_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VU_FF2=$v', d['value'], d['ms_per_step'])"
done; done
