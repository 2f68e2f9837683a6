# Round quantisation of the level-2 sweeps: 13 workgroups per image on 512 (two per CU) or 768 slots.  Per-image time of each sweep at
# batches that fill one round (39: 507 workgroups), 1.3 (52: 676), 1.625 (64: 832) and two rounds (78: 1014).
cd $GRAFT_REPO_ROOT
for b in 39 52 64 78; do
  timeout -k 10 200 python tools/step_tags.py --batch $b --grep "flash" 2>&1 | grep -E "flash" | awk -v b=$b '{printf "B=%d %-40s %8.2f us/launch  %6.3f us/launch/image\n", b, $4, $7, $7/b}'
done
