cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06m_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3"; tail -2 gpurun_out/r06m_tests.log
for B in 64 16; do run 200 gpurun_out/r06m_cb.log python tools/conv_bench.py --B $B --reps 100; grep "C=3" gpurun_out/r06m_cb.log | grep "s= 8\|s=16" | cut -c1-140; done
run 900 gpurun_out/r06m_tf.log python -m pytest tests/test_a_hotpath_gpu.py tests/test_gpu_parity_full.py -x -q; tail -2 gpurun_out/r06m_tf.log
for i in 1 2; do
run 300 gpurun_out/r06m_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06m_bench.log | cut -c60-230
VU_CONV_TZ=0 VU_ATTN_CENTERED_SMALL=0 run 300 gpurun_out/r06m_bench_old.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3 --no-roofline; tail -1 gpurun_out/r06m_bench_old.log | cut -c60-230
done
