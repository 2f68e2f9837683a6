set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/ln_time.py 64 2>&1 | grep "lib="
VU_LN_CHUNK8K=0 timeout -k 10 120 python tools/ln_time.py 64 2>&1 | grep "lib="
timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "layernorm or layer_norm or ln_ or teacher_forced" 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
  VU_LN_CHUNK8K=$v timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline --no-host-input --no-roofline > gpurun_out/ln_bench_$v.log 2>&1 && tail -1 gpurun_out/ln_bench_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CHUNK8K=$v', d['value'], d['ms_per_step'])"
done; done
