# SQ / cache counters of the big-tile GEMM against the vendor kernel on the same product (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ord in 0 1; do
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  VU_BGEMM_ORDER=$ord timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bg_${ord}_$i -- python3 $R/tools/bgemm_probe.py > $R/gpurun_out/pmc_bg_${ord}_$i.log 2>&1
done
done
cd $R
python - <<'PY'
import csv, glob, collections
for ord in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(f"gpurun_out/pmc_bg_{ord}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "bgemm" not in k and "Cijk" not in k: continue
            k = k[:64]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    print(f"== VU_BGEMM_ORDER={ord}: per-launch averages")
    for k, v in agg.items():
        print(k)
        print("   " + "  ".join(f"{n}={v[n] / cnt[k][n]:.4g}" for n in sorted(v)))
PY
rm -rf gpurun_out/pmc_bg_*/
