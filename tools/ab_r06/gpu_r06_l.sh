cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06l_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3"; tail -2 gpurun_out/r06l_tests.log
for NS in 0 340 452 680; do for B in 64 16; do
echo "== NS=$NS B=$B"; VU_TZ_NS=$NS run 200 gpurun_out/r06l_cb.log python tools/conv_bench.py --B $B --reps 100; grep "C=3" gpurun_out/r06l_cb.log | grep "s= 8\|s=16" | cut -c1-140
done; done
VU_TF_DEBUG=1 run 400 gpurun_out/r06l_tf.log python -m pytest tests/test_gpu_parity_full.py -q -s -k "test_teacher_forced_blocks_bf16_full_size and base-dt0-1"; grep "^Decoders.2\|passed\|failed" gpurun_out/r06l_tf.log | cut -c1-60
