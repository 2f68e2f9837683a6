cd $GRAFT_REPO_ROOT
for B in 64 32 16; do for KU in 4 8 4 8; do
echo "KU=$KU"; VU_LN_KU=$KU timeout -k 10 120 python tools/ln_time.py $B 2>&1 | grep -v "^$" | grep "bwd\|fwd"
done; done
