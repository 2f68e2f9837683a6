# which kernels surround the small __amd_rocclr_copyBuffer launches of a step?  (rocprofv3 kernel trace of a short eager run)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rocprof_hunt
rm -rf $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-input --no-graph --no-roofline --no-sustained > $GRAFT_REPO_ROOT/gpurun_out/r05_hunt.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
ev = []
for f in glob.glob("gpurun_out/rocprof_hunt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Queue_Id", "")))
ev.sort()
ev = ev[len(ev) * 2 // 3:]          # the last step
ctx = collections.Counter()
for i, e in enumerate(ev):
    if "copyBuffer" in e[1] or "fillBuffer" in e[1]:
        prev = ev[i - 1][1] if i else ""
        nxt = ev[i + 1][1] if i + 1 < len(ev) else ""
        ctx[(e[1][:24], e[2], prev[:50], nxt[:50])] += 1
for k, v in ctx.most_common(30):
    print(v, k)
PY
rm -rf gpurun_out/rocprof_hunt
