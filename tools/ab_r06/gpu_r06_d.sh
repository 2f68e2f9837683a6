cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
run 400 gpurun_out/r06d_tests.log python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3 or layernorm or ln"; tail -2 gpurun_out/r06d_tests.log
for B in 64 32 16; do
  for SL in 8 16; do VU_LN_BSL=$SL run 100 gpurun_out/r06d_ln_${B}_$SL.log python tools/ln_time.py $B; echo "BSL=$SL"; grep "us per call" gpurun_out/r06d_ln_${B}_$SL.log; done
done
run 200 gpurun_out/r06d_convbench.log python tools/conv_bench.py --B 32; grep "C=3" gpurun_out/r06d_convbench.log | cut -c1-150
run 300 gpurun_out/r06d_bench.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3; tail -1 gpurun_out/r06d_bench.log | cut -c1-330
VU_CONV_TZ=0 VU_LN_BSL=8 run 300 gpurun_out/r06d_bench_old.log python bench.py --no-cpu-baseline --no-host-input --sustained-s 3; tail -1 gpurun_out/r06d_bench_old.log | cut -c1-330
run 300 gpurun_out/r06d_bench16.log python bench.py --batch 16 --no-cpu-baseline --no-host-input --sustained-s 3; tail -1 gpurun_out/r06d_bench16.log | cut -c1-330
VU_CONV_TZ=0 VU_LN_BSL=8 run 300 gpurun_out/r06d_bench16_old.log python bench.py --batch 16 --no-cpu-baseline --no-host-input --sustained-s 3; tail -1 gpurun_out/r06d_bench16_old.log | cut -c1-330
