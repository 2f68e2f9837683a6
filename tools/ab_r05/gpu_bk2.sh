cd $GRAFT_REPO_ROOT
for b in 16 64; do for v in 64 128 256; do
  VU_GEMM_3264_BK=$v timeout -k 10 200 python tools/step_tags.py --batch $b --grep "32x64" 2>&1 | grep -E "32x64" | sed "s/^/BK=$v /"
done; done
