# Round-6 closing checks on the GPU box: bit-reproducibility of the step under GPU sharing for every configuration of the bench
# (tools/nondet_check.py --load), then the training-sanity runs.  Everything lands in gpurun_out/r06_determinism_under_load.txt /
# r06_train_sanity.txt.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_determinism_under_load.txt
echo "# tools/nondet_check.py --load 45 on the round-6 build (probability cache on: forward + loss + backward of one step, repeated; every repetition compared with the first, bit for bit, while a second process trains on the same GPU)" > $O
for cfg in "base 20" "base 64" "lite 32" "large 16"; do
  set -- $cfg
  echo "== $1 B=$2" >> $O
  timeout -k 10 400 python tools/nondet_check.py --model $1 --B $2 --reps 16 --load 45 2>&1 | grep -v amdgpu.ids | grep "rep 1:\|rep 8:\|rep 15:\|load\|NONDET" >> $O
done
cat $O
T=gpurun_out/r06_train_sanity.txt
echo "# tools/train_sanity.py on the round-6 build (fused bf16 step on a fixed synthetic batch, lr 1e-3): last two reports of each run" > $T
timeout -k 10 300 python tools/train_sanity.py base 300 64 2>&1 | grep "step" | tail -2 >> $T
timeout -k 10 300 python tools/train_sanity.py base 300 16 2>&1 | grep "step" | tail -2 >> $T
timeout -k 10 300 python tools/train_sanity.py large 150 16 2>&1 | grep "step" | tail -2 >> $T
timeout -k 10 300 python tools/train_sanity.py seg512 100 32 2>&1 | grep "step" | tail -2 >> $T
timeout -k 10 300 python tools/train_sanity.py lite 200 32 2>&1 | grep "step" | tail -2 >> $T
cat $T
