set -x
cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests -m gpu -q --tb=line > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
# multi-process code path on one GPU (world size 1 over RCCL)
VU_DP_FORCE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-input > gpurun_out/bench_dist1.log 2>&1; tail -c 400 gpurun_out/bench_dist1.log
# rocprofv3 kernel trace of the default bench command (eager for per-kernel rows)
rm -rf $GRAFT_REPO_ROOT/gpurun_out/rocprof_r01; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rocprof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-input --no-graph > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1; tail -c 300 $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log
cd $GRAFT_REPO_ROOT; find gpurun_out/rocprof_r01 -name "*stats*" | head; du -sh gpurun_out/rocprof_r01
# keep only the stats csv (kernel trace itself is large)
find gpurun_out/rocprof_r01 -name "*kernel_trace.csv" -size +20M -delete
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.log 2>&1; tail -c 1500 gpurun_out/bench_full.log
