cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 600 gpurun_out/r06r_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "flash"; tail -2 gpurun_out/r06r_tests.log
for i in 1 2; do
run 300 gpurun_out/r06r_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06r_bench.log | cut -c60-230
done
run 300 gpurun_out/r06r_b16.log python bench.py --batch 16 --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06r_b16.log | cut -c60-230
run 300 gpurun_out/r06r_l16.log python bench.py --model large --batch 16 --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06r_l16.log | cut -c60-230
run 300 gpurun_out/r06r_lite.log python bench.py --model lite --batch 32 --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06r_lite.log | cut -c60-230
