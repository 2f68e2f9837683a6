cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/rocprof_tmp; rm -rf $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-input --no-graph --no-roofline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/rocprof_tmp/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"][:60] for r in rows]
idx=[i for i,n in enumerate(names) if "copyBuffer" in n]
print(len(rows), "launches", len(idx), "copies")
seen=set()
for i in idx[-60:]:
    ctx=(names[i-1] if i>0 else "", names[i+1] if i+1<len(names) else "")
    if ctx in seen: continue
    seen.add(ctx); print("before:", ctx[0], "| after:", ctx[1], "| grid", rows[i].get("Grid_Size_X", rows[i].get("Grid_Size")))
import shutil; shutil.rmtree("gpurun_out/rocprof_tmp", ignore_errors=True)
PY
