cd $GRAFT_REPO_ROOT
for b in 16 64; do for v in 64 128 256 257; do
  VU_GEMM_3232_BK=$v timeout -k 10 200 python tools/step_tags.py --batch $b --grep "32x32" 2>&1 | grep -E "32x32" | sed "s/^/BK=$v /"
done; done
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
for v in 128 256; do VU_GEMM_3232_BK=$v timeout -k 10 300 python -m pytest tests/test_gpu_parity_full.py tests/test_a_hotpath_gpu.py -x -q -m gpu -k "teacher_forced or feedforward" 2>&1 | tail -1; done
