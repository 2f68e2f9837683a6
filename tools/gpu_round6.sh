# Round-6 record run (on the GPU box: bash tools/gpu_round6.sh [part]): bench lines of every BASELINE configuration, a kernel
# trace, the HBM counters, MFMA busy and the SQ counters of the recompute sweeps.  Everything lands in gpurun_out/r06_*;
# tools/record_round6.py copies the summaries into profiles/.  Parts: bench | trace | pmc | all (default).
set -x
PART=${1:-all}
cd $GRAFT_REPO_ROOT
if [ "$PART" = bench ] || [ "$PART" = all ]; then
  python bench.py > gpurun_out/r06_bench_base.log 2>&1; tail -1 gpurun_out/r06_bench_base.log > gpurun_out/r06_bench_base.json; cut -c1-300 gpurun_out/r06_bench_base.json
  for cfg in "lite 32" "large 16" "seg512 32" "seg512 8" "base 16" "base 32" "base 128"; do
    set -- $cfg
    timeout 600 python bench.py --model $1 --batch $2 --no-cpu-baseline --no-host-input > gpurun_out/r06_bench_$1_$2.log 2>&1
    tail -1 gpurun_out/r06_bench_$1_$2.log > gpurun_out/r06_bench_$1_$2.json; cut -c1-200 gpurun_out/r06_bench_$1_$2.json; echo
  done
  timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-host-input > gpurun_out/r06_bench_base_fp32.log 2>&1
  tail -1 gpurun_out/r06_bench_base_fp32.log > gpurun_out/r06_bench_base_fp32.json; cut -c1-200 gpurun_out/r06_bench_base_fp32.json; echo
  timeout 600 python bench.py --model seg512 --batch 32 --attn-operands storage --no-cpu-baseline --no-host-input > gpurun_out/r06_bench_seg512_32_storage.log 2>&1
  tail -1 gpurun_out/r06_bench_seg512_32_storage.log > gpurun_out/r06_bench_seg512_32_storage.json; cut -c1-200 gpurun_out/r06_bench_seg512_32_storage.json; echo
fi
if [ "$PART" = trace ] || [ "$PART" = all ]; then
  bash tools/gpu_trace.sh r06 > gpurun_out/r06_trace.log 2>&1; tail -3 gpurun_out/r06_trace.log
fi
if [ "$PART" = pmc0 ]; then      # only the HBM counters of the default workload
  bash tools/gpu_pmc.sh > gpurun_out/r06_pmc.log 2>&1; tail -12 gpurun_out/r06_pmc.log
  for C in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_${C}_summary.csv gpurun_out/r06_base_64_pmc_${C}_summary.csv; done
fi
if [ "$PART" = pmc ] || [ "$PART" = all ]; then
  bash tools/gpu_pmc.sh > gpurun_out/r06_pmc.log 2>&1; tail -12 gpurun_out/r06_pmc.log
  for C in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_${C}_summary.csv gpurun_out/r06_base_64_pmc_${C}_summary.csv; done      # (pmc2 below reuses the file names)
  bash tools/gpu_pmc_mfma.sh > gpurun_out/r06_pmc_mfma.log 2>&1; tail -12 gpurun_out/r06_pmc_mfma.log
  bash tools/gpu_pmc_flash.sh r06_flash > gpurun_out/r06_pmc_flash.log 2>&1; tail -5 gpurun_out/r06_pmc_flash.log
fi
# HBM counters of the dominant kernels of the other BASELINE configurations (round-4 review: `traffic` was null for them): the flash
# family only, summaries kept per workload for tools/record_round6.py
if [ "$PART" = pmc2 ] || [ "$PART" = all ]; then
  for cfg in "lite 32" "base 16" "large 16"; do
    set -- $cfg
    BENCH_ARGS="--model $1 --batch $2" PMC_FAMILIES="flash" bash tools/gpu_pmc.sh > gpurun_out/r06_pmc_$1_$2.log 2>&1
    for C in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_${C}_summary.csv gpurun_out/r06_$1_$2_pmc_${C}_summary.csv; done
    tail -4 gpurun_out/r06_pmc_$1_$2.log
  done
fi
