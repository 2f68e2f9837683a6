#!/bin/bash
# Round 4, VERDICT item 1: where does the two-rank Base bf16 rehearsal lose bit-identity?  Every leg writes its own log.
set -o pipefail
O=gpurun_out/bisect; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 280 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(grep -h 'NONDET_CHECK\|bit for bit\|forward output\|CONTENTION' $O/$name.log | tr '\n' ' ' | cut -c1-400)"; }
N="python tools/nondet_check.py --B 20"
run wsd          $N --load 40 --ws-diff 30
run lfp          $N --reps 8 --load 40 --fresh --poison --twice
run ops          python tools/contention_ops.py --load 50 --iters 150
run reh1  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/dp_rehearsal.py --base
run reh2  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 tools/dp_rehearsal.py --base
VU_CONV_W=smem run reh_smem  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 tools/dp_rehearsal.py --base
echo "== conv bench (LDS weights, then scalar weights)"
python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids
VU_CONV_W=smem python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids
echo BISECT_DONE
