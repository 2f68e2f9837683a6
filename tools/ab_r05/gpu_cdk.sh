cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "flash" 2>&1 | tail -2
for b in 16 64; do for v in 0 1; do
  VU_CENTER_DK_WIDE=$v timeout -k 10 200 python tools/step_tags.py --batch $b --grep "center" 2>&1 | grep -E "center" | sed "s/^/WIDE=$v /"
done; done
VU_CENTER_DK_WIDE=1 timeout -k 10 200 python tools/step_tags.py --model lite --batch 32 --grep "center" 2>&1 | grep -E "center" | sed "s/^/WIDE=1 /"
VU_CENTER_DK_WIDE=0 timeout -k 10 200 python tools/step_tags.py --model lite --batch 32 --grep "center" 2>&1 | grep -E "center" | sed "s/^/WIDE=0 /"
