set -x
cd $GRAFT_REPO_ROOT
for b in 64 16; do
  timeout -k 10 120 python tools/ln_time.py $b 2>&1 | grep "lib="
  for its in 3 4 6; do VU_LIB_PATH=$PWD/tmp_variants/lib_lnits$its.so timeout -k 10 120 python tools/ln_time.py $b 2>&1 | grep "lib="; done
done
