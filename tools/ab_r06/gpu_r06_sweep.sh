# in-step sweep of the tile / grid switches that were chosen by stand-alone measurements (round 6: the step can disagree)
cd $GRAFT_REPO_ROOT
b() { python bench.py --batch $1 --no-cpu-baseline --no-host-input --no-sustained --no-roofline 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/' | cut -c1-7; }
for B in 64 16; do
echo "B=$B base $(b $B) $(b $B)"
for kv in VU_TZ_NS=340 VU_TZ_NS=1020 VU_TZ_PF=4 VU_LN_CHUNK8K=0 VU_LN_CHUNK8K=1 VU_MAP_BWD_CAP=256 VU_MAP_BWD_CAP=1024 VU_GEMM_QUARTER_BELOW=100 VU_GEMM_QUARTER_BELOW=400 VU_GEMM_HALF_BELOW=200 VU_GEMM_HALF_BELOW=800 VU_TSGEMM_HALVES=0 VU_BN_BWD_WIDE=1 VU_CENTER_DK_WIDE=0 VU_CENTER_DK_WIDE=1 VU_GEMM_3232=0 VU_GEMM_EIGHTH=0 VU_GEMM_HALFROW=0 VU_GEMM_SMALLK=0 VU_PGEMM=0 VU_FF2=0 VU_FLASH_DK3=0 VU_ATTN_F1=0 VU_ATTN_F1=2 VU_BGEMM_SMALL=1 VU_BGEMM_SMALL=0 VU_BGEMM_SHORTK_M=2048 VU_TSGEMM_SPLITS=8 VU_FLASH_KS=1 VU_FLASH_KS=2; do
echo "B=$B $kv $(env $kv bash -c "$(declare -f b); b $B")"
done
echo "B=$B base $(b $B)"
done
