cd $GRAFT_REPO_ROOT
for b in 64 16; do for v in 1 2; do
  VU_PGEMM=$v timeout -k 10 200 python tools/step_tags.py --batch $b --grep "192" 2>&1 | grep -E "N192|K192|192x192>" | grep -v tsgemm | sed "s/^/PGEMM=$v /" | cut -c1-150
done; done
for rep in 1 2; do for v in 1 2; do
  VU_PGEMM=$v timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/pg.log 2>&1 && tail -1 gpurun_out/pg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH PGEMM=$v B=64', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
