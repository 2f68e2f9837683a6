cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 900 gpurun_out/r06fin_tests.log python -m pytest tests/test_gpu_model.py tests/test_a_hotpath_gpu.py -x -q; tail -3 gpurun_out/r06fin_tests.log
for i in 1 2; do for V in 1 0; do for B in 64 16; do
VU_FIN_FUSE=$V run 300 gpurun_out/r06fin_bench.log python bench.py --batch $B --no-cpu-baseline --no-host-input --no-sustained --no-roofline; echo "FIN=$V B=$B $(tail -1 gpurun_out/r06fin_bench.log | cut -c60-200)"
done; done; done
run 300 gpurun_out/r06fin_nd.log python tools/nondet_check.py --B 20 --reps 6 --poison --load 20; tail -4 gpurun_out/r06fin_nd.log
