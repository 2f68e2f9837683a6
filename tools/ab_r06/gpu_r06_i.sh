cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06i_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "centred_map_form"; tail -3 gpurun_out/r06i_tests.log
run 400 gpurun_out/r06i_hot.log python -m pytest tests/test_a_hotpath_gpu.py tests/test_gpu_parity_full.py -x -q -k "hotpath or teacher_forced"; tail -3 gpurun_out/r06i_hot.log
for i in 1 2; do
run 300 gpurun_out/r06i_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06i_bench.log | cut -c60-230
VU_ATTN_CENTERED_SMALL=0 run 300 gpurun_out/r06i_bench_old.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06i_bench_old.log | cut -c60-230
done
run 300 gpurun_out/r06i_b16.log python bench.py --batch 16 --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06i_b16.log | cut -c60-230
VU_ATTN_CENTERED_SMALL=0 run 300 gpurun_out/r06i_b16o.log python bench.py --batch 16 --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06i_b16o.log | cut -c60-230
