cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
for B in 64 16; do
for PF in 4 2; do for NS in 680 340 1020; do
  echo "== B=$B PF=$PF NS=$NS"
  VU_TZ_PF=$PF VU_TZ_NS=$NS run 200 gpurun_out/r06c_$B_$PF_$NS.log python tools/conv_bench.py --B $B --reps 200; grep "s= 8\|s=16" gpurun_out/r06c_$B_$PF_$NS.log | grep "C=3" | cut -c1-150
done; done; done
