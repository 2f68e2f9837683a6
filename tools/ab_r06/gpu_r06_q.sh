cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
for V in base ST AP DL KV all base; do
  if [ $V = base ]; then LIBV=""; else LIBV="VU_LIB_PATH=$GRAFT_REPO_ROOT/tmp_variants/lib_iglp_$V.so"; fi
  env $LIBV timeout -k 10 300 python bench.py --model lite --batch 32 --no-cpu-baseline --no-host-input --no-sustained --no-roofline > gpurun_out/r06q_$V.log 2>&1
  echo "$V: $(tail -1 gpurun_out/r06q_$V.log | cut -c60-200)"
done
