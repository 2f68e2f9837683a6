# Fused level-2 FeedForward (csrc/vu_ff2.hip): parity subset, A/B of the step (VU_FF2=0 / 1), and a by-grid trace of Base 16 images.
set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity_full.py tests/test_a_hotpath_gpu.py -x -q -m gpu -k "feed or linear or teacher_forced or poison or vendor" > gpurun_out/ff2_tests.txt 2>&1 || { tail -30 gpurun_out/ff2_tests.txt; exit 1; }
tail -3 gpurun_out/ff2_tests.txt
for rep in 1 2; do for v in 0 1; do
  VU_FF2=$v timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/ff2_bench_$v.log 2>&1 && tail -1 gpurun_out/ff2_bench_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VU_FF2=$v', d['value'], d['ms_per_step'])"
done; done
VU_FF2=1 timeout -k 10 300 python bench.py --model base --batch 16 --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/ff2_bench_b16.log 2>&1 && tail -1 gpurun_out/ff2_bench_b16.log | cut -c1-160
bash tools/gpu_trace.sh r05b16_base --model base --batch 16 > gpurun_out/r05b16_trace.log 2>&1; tail -3 gpurun_out/r05b16_trace.log
