cd $GRAFT_REPO_ROOT
for cfg in "base 16" "large 16" "base 32"; do set -- $cfg
for v in "2048 22" "2048 21" "512 21"; do set -- $cfg $v
  VU_BGEMM_SHORTK_M=$3 VU_BGEMM_SHORTK_MN=$4 timeout -k 10 300 python bench.py --model $1 --batch $2 --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/sk.log 2>&1 && tail -1 gpurun_out/sk.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SHORTK M>=$3 MN>=2^$4 $1 $2', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
VU_PROF_SHAPES=1 VU_BGEMM_SHORTK_M=512 VU_BGEMM_SHORTK_MN=21 timeout -k 10 200 python tools/step_tags.py --batch 16 --grep "K64 \|K128 " 2>&1 | grep -E "K64|K128" | cut -c30-150
