# Round-2 record run: the default bench line, the other BASELINE configurations, a kernel trace and the HBM counters.
set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02_bench_base.log 2>&1; tail -1 gpurun_out/r02_bench_base.log > gpurun_out/r02_bench_base.json; cut -c1-400 gpurun_out/r02_bench_base.json
for cfg in "lite 32" "large 16" "seg512 32" "seg512 8" "base 16" "base 32" "base 128"; do
  set -- $cfg
  timeout 600 python bench.py --model $1 --batch $2 --no-cpu-baseline --no-host-input > gpurun_out/r02_bench_$1_$2.log 2>&1
  tail -1 gpurun_out/r02_bench_$1_$2.log > gpurun_out/r02_bench_$1_$2.json; cut -c1-260 gpurun_out/r02_bench_$1_$2.json; echo
done
timeout 600 python bench.py --model seg512 --batch 32 --attn-operands storage --no-cpu-baseline --no-host-input > gpurun_out/r02_bench_seg512_32_storage.log 2>&1
tail -1 gpurun_out/r02_bench_seg512_32_storage.log > gpurun_out/r02_bench_seg512_32_storage.json; cut -c1-260 gpurun_out/r02_bench_seg512_32_storage.json; echo
bash tools/gpu_trace.sh r02d > gpurun_out/r02d_trace.log 2>&1; tail -3 gpurun_out/r02d_trace.log
bash tools/gpu_pmc.sh > gpurun_out/r02_pmc.log 2>&1; tail -30 gpurun_out/r02_pmc.log
