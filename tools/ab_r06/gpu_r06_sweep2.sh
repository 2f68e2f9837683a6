cd $GRAFT_REPO_ROOT
b() { python bench.py --model $1 --batch $2 --no-cpu-baseline --no-host-input --no-sustained --no-roofline 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/' | cut -c1-7; }
for cfg in "lite 32" "seg512 32" "seg512 8" "large 16"; do set -- $cfg
echo "$1 $2 base $(b $1 $2) $(b $1 $2)"
for kv in VU_BGEMM_T64=0 VU_BGEMM_TILE=0 VU_BGEMM_TILE=1 VU_BGEMM_TILE=3 VU_GEMM_F32_QUARTER=0 VU_DEFER_RED=0 VU_GEMM_HALF_BELOW=200 VU_GEMM_QUARTER_BELOW=100 VU_TZ_NS=340 VU_BGEMM_SMALL=1; do
echo "$1 $2 $kv $(env $kv bash -c "$(declare -f b); b $1 $2")"
done; done
