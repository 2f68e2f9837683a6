cd $GRAFT_REPO_ROOT
for V in 0 2 3 0 2 3; do echo "T64=$V"; VU_BGEMM_T64=$V timeout -k 10 200 python tools/gemm_small_batch.py 2>&1 | grep "M784\|M1568"; done
