cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
for i in 1 2; do for V in 0 2 1; do for B in 64 16; do
VU_SIDE_LANE=$V run 300 gpurun_out/r06v_bench.log python bench.py --batch $B --no-cpu-baseline --no-host-input --no-sustained --no-roofline; echo "LANE=$V B=$B $(tail -1 gpurun_out/r06v_bench.log | cut -c60-200)"
done; done; done
