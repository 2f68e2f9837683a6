cd $GRAFT_REPO_ROOT
b() { python bench.py --model $1 --batch $2 --no-cpu-baseline --no-host-input --no-sustained --no-roofline 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/' | cut -c1-7; }
for cfg in "lite 32" "seg512 32"; do set -- $cfg
echo "$1 $2 base $(b $1 $2) $(b $1 $2)"
for kv in VU_FLASH_KS=1 VU_FLASH_KS=2 VU_FLASH_FUSE_DQ=0 VU_FLASH_FORK_V1=1 VU_FLASH_FORK_V1=0 VU_FLASH_FORK=1 VU_LN_CHUNK8K=0 VU_LN_CHUNK8K=1 VU_TZ_NS=1020 VU_TZ_PF=4 VU_PGEMM=0 VU_FF2=0 VU_GEMM_3232=0 VU_GEMM_EIGHTH=0 VU_MAP_BWD_CAP=1024 VU_FLASH_PCACHE=0 VU_ATTN_CENTERED=0; do
echo "$1 $2 $kv $(env $kv bash -c "$(declare -f b); b $1 $2")"
done; done
