cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 300 gpurun_out/r06b_convtest.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3"; tail -3 gpurun_out/r06b_convtest.log
for B in 64 16; do
  run 200 gpurun_out/r06b_convbench_tz_$B.log python tools/conv_bench.py --B $B; grep "C=3" gpurun_out/r06b_convbench_tz_$B.log
  VU_CONV_TZ=0 run 200 gpurun_out/r06b_convbench_old_$B.log python tools/conv_bench.py --B $B; grep "C=3" gpurun_out/r06b_convbench_old_$B.log
done
for S in 16 8; do
  VU_CONV_TZ=0 VU_CONV_W=smem run 200 gpurun_out/r06b_alias_$S.log python tools/sharing_alias.py --secs 20 --s $S; grep "child" gpurun_out/r06b_alias_$S.log
done
