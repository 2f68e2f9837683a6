# rocprofv3 kernel trace of one bench.py run (eager launches), reduced to two summaries under gpurun_out/:
#   <tag>_kernel_stats.csv   rocprofv3's own --stats table
#   <tag>_by_grid.csv        per (kernel, grid size): launches per step, average and total us per step
# usage: bash tools/gpu_trace.sh <tag> [bench.py args...]
set -x
TAG=${1:-r02}; shift
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG
rm -rf $OUT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-input --no-graph --no-roofline --no-sustained "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.log 2>&1
tail -c 300 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.log
cd $GRAFT_REPO_ROOT
python - "$TAG" <<'PY'
import csv, glob, collections, shutil, sys
tag = sys.argv[1]
root = f"gpurun_out/rocprof_{tag}"
for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"gpurun_out/{tag}_kernel_stats.csv")
agg = collections.defaultdict(lambda: [0, 0.0])
steps = 6
# the queue a kernel ran on: the main compute stream is the queue with most launches; every other queue is a forked stream - the
# low-priority stream of the recompute backward's dv sweep (csrc/vu_flash.hip "Tail overlap"), whose WALL time in this table
# includes the time its workgroups waited for slots the dq / dk sweeps held (the same kernel alone: profiles/*_flash_alone.txt)
qcount = collections.Counter()
recs = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        qid = r.get("Queue_Id", r.get("Stream_Id", ""))
        qcount[qid] += 1
        recs.append((r, qid))
main_q = qcount.most_common(1)[0][0] if qcount else ""
# Only whole steps count (round 6): everything before the END of the first step's optimizer launch is dropped - the per-parameter
# copies of model.to(device) at construction (the "48 copyBuffer launches per step" of the round-5 table were these ~290 launches
# divided by 6 steps; there is no device-to-device copy inside a step) and the first warm-up step - and the rest is divided by the
# number of steps that follow.
marks = sorted(int(r["End_Timestamp"]) for r, _ in recs if r["Kernel_Name"].startswith("adamw_step_kernel"))
if not marks:
    marks = sorted(int(r["End_Timestamp"]) for r, _ in recs if r["Kernel_Name"].startswith("adamw_kernel"))
if len(marks) >= 2:
    recs = [(r, q) for r, q in recs if marks[0] <= int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= marks[-1]]
    steps = len(marks) - 1
for r, qid in recs:
    k = (r["Kernel_Name"][:110], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), "main" if qid == main_q else "side (low priority)")
    a = agg[k]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = sorted(agg.items(), key=lambda kv: (kv[0][4] != "main", -kv[1][1]))      # the main stream first: its top row is the dominant kernel
tot = sum(v[1] for _, v in rows)
with open(f"gpurun_out/{tag}_by_grid.csv", "w") as o:
    o.write("kernel,grid_x,grid_y,grid_z,stream,launches_per_step,avg_us,us_per_step,share\n")
    for (k, gx, gy, gz, q), (n, us) in rows[:140]:
        o.write('"%s",%s,%s,%s,%s,%.1f,%.1f,%.1f,%.4f\n' % (k, gx, gy, gz, q, n / steps, us / n, us / steps, us / tot))
print("total us per step (all launches / %d steps): %.1f" % (steps, tot / steps))
# idle time between consecutive kernels of the compute stream (end of one -> start of the next), second half of the trace
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
ev.sort()
ev = ev[len(ev) // 2:]
gaps = sorted((b[0] - a[1]) / 1e3 for a, b in zip(ev, ev[1:]))
small = [g for g in gaps if g < 200.0]
busy = sum((e[1] - e[0]) for e in ev) / 1e3
with open(f"gpurun_out/{tag}_gaps.txt", "w") as o:
    o.write("launches %d  busy %.1f us  gaps(<200us) %.1f us  = %.1f %% of busy+gaps\n" % (len(ev), busy, sum(small), 100.0 * sum(small) / (busy + sum(small))))
    o.write("gap us: p10 %.2f  median %.2f  p90 %.2f  max(<200) %.2f  overlapping (<0): %d\n" % (gaps[len(gaps) // 10], gaps[len(gaps) // 2], gaps[9 * len(gaps) // 10], max(small), sum(1 for g in gaps if g < 0)))
# the same for the MAIN stream alone: time its queue sat idle between two of its own kernels (waiting for a forked stream to join,
# or for the host), second half of the trace
mev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r, qid in recs if qid == main_q)
mev = mev[len(mev) // 2:]
idle = [((b[0] - a[1]) / 1e3, a[2], b[2]) for a, b in zip(mev, mev[1:])]
small = [g for g in idle if 0.0 < g[0] < 2000.0]
span = (mev[-1][1] - mev[0][0]) / 1e3
with open(f"gpurun_out/{tag}_gaps.txt", "a") as o:
    o.write("main stream alone: %d launches over %.1f us; idle between its kernels %.1f us = %.1f %% (gaps > 3 us: %d, sum %.1f us)\n" % (
        len(mev), span, sum(g[0] for g in small), 100.0 * sum(g[0] for g in small) / span, sum(1 for g in small if g[0] > 3.0), sum(g[0] for g in small if g[0] > 3.0)))
    for g in sorted(small, reverse=True)[:12]:
        o.write("  %.1f us idle after %s before %s\n" % g)
print(open(f"gpurun_out/{tag}_gaps.txt").read())
shutil.rmtree(root, ignore_errors=True)
PY
head -45 gpurun_out/${TAG}_by_grid.csv
