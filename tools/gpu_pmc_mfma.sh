# MFMA utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs) against
# GRBM_GUI_ACTIVE (sum over the 8 XCDs) - one --pmc pass, kernel-trace only.
cd /tmp && export TMPDIR=/tmp
# (one pass per kernel family: see tools/gpu_pmc.sh)
FAMILIES=('flash' 'gemm' 'ln_|layernorm' 'map_|scores|mix_' 'conv' 'adamw|retile|bn_|colsum|cast|mse|tsgemm')
i=0
for RX in "${FAMILIES[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma/f$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-input --no-graph --no-roofline --no-sustained > $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma_f$i.log 2>&1
  echo "mfma / $RX: rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:80]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:24]
names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_WAVES"]
with open("gpurun_out/pmc_mfma_summary.csv", "w") as o:
    # mfma_util = MFMA busy cycles / (SIMDs * kernel cycles); 256 CUs * 4 SIMDs; GRBM_GUI_ACTIVE is summed over 8 XCDs
    o.write("kernel,launches," + ",".join(n + "_per_launch" for n in names) + ",mfma_util\n")
    for k, v in rows:
        n = max(cnt[k], 1)
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
        util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc) if cyc else 0.0
        o.write('"%s",%d,' % (k, cnt[k]) + ",".join("%.0f" % (v.get(x, 0) / n) for x in names) + ",%.4f\n" % util)
print(open("gpurun_out/pmc_mfma_summary.csv").read())
import shutil; shutil.rmtree("gpurun_out/pmc_mfma", ignore_errors=True)
PY
