cd $GRAFT_REPO_ROOT
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "KILLED: $*"; tail -5 $log; exit 1; fi; return 0; }
for V in "VU_CONV_TZ=1 VU_ATTN_CENTERED_SMALL=1" "VU_CONV_TZ=0 VU_ATTN_CENTERED_SMALL=1" "VU_CONV_TZ=1 VU_ATTN_CENTERED_SMALL=0" "VU_CONV_TZ=0 VU_ATTN_CENTERED_SMALL=0"; do
  echo "== $V"
  env $V VU_TF_DEBUG=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity_full.py -q -s -k "test_teacher_forced_blocks_bf16_full_size and base-dt0-1" > gpurun_out/r06j.log 2>&1
  grep "^Decoders\|^Encoders\|^BottleNeck\|passed\|failed" gpurun_out/r06j.log | cut -c1-60
done
