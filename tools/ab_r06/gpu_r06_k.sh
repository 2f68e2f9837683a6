cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06k_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3 or centred_map_form"; tail -2 gpurun_out/r06k_tests.log
run 200 gpurun_out/r06k_cb.log python tools/conv_bench.py --B 64 --reps 100; grep "C=3" gpurun_out/r06k_cb.log | cut -c60-260
run 200 gpurun_out/r06k_cb16.log python tools/conv_bench.py --B 16 --reps 100; grep "C=3" gpurun_out/r06k_cb16.log | cut -c60-260
run 900 gpurun_out/r06k_tf.log python -m pytest tests/test_a_hotpath_gpu.py tests/test_gpu_parity_full.py -x -q; tail -3 gpurun_out/r06k_tf.log
run 300 gpurun_out/r06k_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06k_bench.log | cut -c60-230
