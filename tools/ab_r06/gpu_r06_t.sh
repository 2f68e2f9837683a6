cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06t_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3"; tail -2 gpurun_out/r06t_tests.log
for B in 64 16; do run 200 gpurun_out/r06t_cb.log python tools/conv_bench.py --B $B --reps 100; grep "C=3" gpurun_out/r06t_cb.log | grep "s= 8\|s=16" | cut -c50-250; done
PMC_FAMILIES="conv" bash tools/gpu_pmc.sh > gpurun_out/r06t_pmc.log 2>&1; grep "conv_tz" gpurun_out/pmc_FETCH_SIZE_summary.csv gpurun_out/pmc_WRITE_SIZE_summary.csv | cut -c1-200
for i in 1 2; do run 300 gpurun_out/r06t_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06t_bench.log | cut -c60-230; done
run 300 gpurun_out/r06t_b16.log python bench.py --batch 16 --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06t_b16.log | cut -c60-230
