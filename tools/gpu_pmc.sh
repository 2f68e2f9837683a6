# HBM traffic counters for the dominant kernels (separate --pmc passes, no other tracing domains)
set -x
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
# (round 3: one pass per kernel family - with the counters armed on every dispatch of a step this image's rocprofv3 dies with
# a SIGSEGV in its own dispatch callback; each family alone, 300 - 900 dispatches, runs through)
# BENCH_ARGS: extra bench.py arguments (e.g. "--model lite --batch 32": the PMC passes of another BASELINE configuration);
# PMC_FAMILIES="flash": only these kernel families (the other configurations only need their dominant kernels)
FAMILIES=('flash' 'gemm' 'ln_|layernorm' 'map_|scores|mix_' 'conv' 'adamw|retile|bn_|colsum|cast|mse|tsgemm')
if [ -n "$PMC_FAMILIES" ]; then IFS=' ' read -r -a FAMILIES <<< "$PMC_FAMILIES"; fi
for C in FETCH_SIZE WRITE_SIZE; do
  i=0
  for RX in "${FAMILIES[@]}"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C/f$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-input --no-graph --no-roofline --no-sustained $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_${C}_f$i.log 2>&1
    echo "$C / $RX: rc=$?"
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/pmc_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                k = r["Kernel_Name"][:160]
                agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])      # every symbol: the two passes are joined on the name (pmc_traffic_json.py)
    with open(f"gpurun_out/pmc_{C}_summary.csv", "w") as o:
        o.write("kernel,launches,sum_%s,avg_per_launch\n" % C)
        for k, (n, v) in rows:
            o.write('"%s",%d,%.1f,%.1f\n' % (k, n, v, v / n))
        for k, (n, v) in rows[:24]:
            print(C, n, "%.1f" % (v / n), k[:90])
    # drop the bulky raw files
import shutil
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.rmtree(f"gpurun_out/pmc_{C}", ignore_errors=True)
PY
