cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06h_hot.log python -m pytest tests/test_a_hotpath_gpu.py -x -q; tail -2 gpurun_out/r06h_hot.log
for i in 1 2; do
run 300 gpurun_out/r06h_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06h_bench.log | cut -c1-230
VU_CONV_TZ=0 run 300 gpurun_out/r06h_bench_old.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06h_bench_old.log | cut -c1-230
done
for cfg in "base 16" "large 16" "lite 32"; do set -- $cfg
run 300 gpurun_out/r06h_b_$1_$2.log python bench.py --model $1 --batch $2 --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06h_b_$1_$2.log | cut -c1-230
VU_CONV_TZ=0 run 300 gpurun_out/r06h_bo_$1_$2.log python bench.py --model $1 --batch $2 --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06h_bo_$1_$2.log | cut -c1-230
done
