cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06s_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3"; tail -2 gpurun_out/r06s_tests.log
for NS in 0 170 226 340; do for B in 64 16; do
echo "== NS=$NS B=$B"; VU_TZ_NS=$NS run 200 gpurun_out/r06s_cb.log python tools/conv_bench.py --B $B --reps 100; grep "C=3" gpurun_out/r06s_cb.log | grep "s= 8\|s=16" | cut -c55-110
done; done
run 300 gpurun_out/r06s_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06s_bench.log | cut -c60-230
