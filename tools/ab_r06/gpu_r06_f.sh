cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06f_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3"; tail -3 gpurun_out/r06f_tests.log
for B in 64 16; do
run 200 gpurun_out/r06f_cb_$B.log python tools/conv_bench.py --B $B --reps 100; grep "C=3" gpurun_out/r06f_cb_$B.log | cut -c60-260
VU_CONV_TZ=0 run 200 gpurun_out/r06f_cbo_$B.log python tools/conv_bench.py --B $B --reps 100; grep "C=3" gpurun_out/r06f_cbo_$B.log | cut -c60-260
done
run 300 gpurun_out/r06f_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06f_bench.log | cut -c1-230
VU_CONV_TZ=0 run 300 gpurun_out/r06f_bench_old.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06f_bench_old.log | cut -c1-230
