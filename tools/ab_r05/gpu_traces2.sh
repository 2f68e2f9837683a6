set -x
cd $GRAFT_REPO_ROOT
bash tools/gpu_trace.sh r05b16_large --model large --batch 16 > gpurun_out/r05b16_large_trace.log 2>&1; tail -3 gpurun_out/r05b16_large_trace.log
bash tools/gpu_trace.sh r05b32_lite --model lite --batch 32 > gpurun_out/r05b32_lite_trace.log 2>&1; tail -3 gpurun_out/r05b32_lite_trace.log
