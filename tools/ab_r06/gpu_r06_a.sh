# Round 6, first GPU call: sanity of the build, the default bench line, the fp32 bench line (review item 7 iii), and the sharing
# experiment of review item 6 (the two FAILING builds kept from round 4: scalar-load conv weights, LDS-row mix table - beside a
# second PROCESS vs beside a second STREAM of the same process).  A step that is killed at its limit ends the call.
cd $GRAFT_REPO_ROOT
run() {  # run <limit s> <log> <cmd...>: stop the whole call when the step had to be killed
  local lim=$1 log=$2; shift 2
  timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED at its limit: $*"; tail -5 $log; exit 1; fi
  return 0
}
run 400 gpurun_out/r06a_hotpath.log python -m pytest tests/test_a_hotpath_gpu.py -x -q; tail -2 gpurun_out/r06a_hotpath.log
run 500 gpurun_out/r06a_bench_base.log python bench.py; tail -1 gpurun_out/r06a_bench_base.log > gpurun_out/r06a_bench_base.json; cut -c1-400 gpurun_out/r06a_bench_base.json; echo
run 400 gpurun_out/r06_bench_base_fp32.log python bench.py --dtype fp32 --no-cpu-baseline --no-host-input; tail -1 gpurun_out/r06_bench_base_fp32.log > gpurun_out/r06_bench_base_fp32.json; cut -c1-300 gpurun_out/r06_bench_base_fp32.json; echo
O=gpurun_out/r06_sharing_experiment.txt
echo "# round 6, review item 6: do the two sharing faults need a second PROCESS?  tools/contention_ops.py, failing builds" > $O
for MODE in process thread; do
  echo "== conv_fwd with scalar-load weights (VU_CONV_W=smem), load = a Base train-step loop as a second $MODE" >> $O
  VU_CONV_W=smem run 300 gpurun_out/r06_share_conv_$MODE.log python tools/contention_ops.py --load 45 --iters 120 --only conv --attn none --load-mode $MODE
  grep "load\|repetitions\|CONTENTION" gpurun_out/r06_share_conv_$MODE.log | grep -v amdgpu.ids >> $O
done
for MODE in process thread; do
  echo "== 4-head recompute backward with the mix table rows prefetched from LDS (-DVU_V1_FW_LDS=1), load = second $MODE" >> $O
  CONTENTION_ATTN_ONLY=1 VU_LIB_PATH=$GRAFT_REPO_ROOT/tmp_variants/lib_v1fwlds.so run 400 gpurun_out/r06_share_v1_$MODE.log python tools/contention_ops.py --load 60 --iters 40 --attn all --load-mode $MODE
  grep "load\|repetitions\|CONTENTION" gpurun_out/r06_share_v1_$MODE.log | grep -v amdgpu.ids >> $O
done
echo "== control: the product build (LDS weights, register table), second process" >> $O
run 300 gpurun_out/r06_share_control.log python tools/contention_ops.py --load 40 --iters 60 --only conv --attn all
grep "load\|repetitions\|CONTENTION" gpurun_out/r06_share_control.log | grep -v amdgpu.ids >> $O
cat $O
