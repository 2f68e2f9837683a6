# probability-cache A/B on the GPU box: the default build on five shapes, then measurement builds on the Base shape
cd $GRAFT_REPO_ROOT
python tools/pcache_ab.py > gpurun_out/r05_pc_ab64.txt 2>&1
python tools/pcache_ab.py --B 16 > gpurun_out/r05_pc_ab16.txt 2>&1
python tools/pcache_ab.py --B 8 --N 1024 --C 1 --s 16 > gpurun_out/r05_pc_ab_d32.txt 2>&1
python tools/pcache_ab.py --B 2 --N 4096 --C 1 --s 8 > gpurun_out/r05_pc_ab_d8.txt 2>&1
python tools/pcache_ab.py --B 20 --ks 2 --cross > gpurun_out/r05_pc_ab_ks2.txt 2>&1
python tools/pcache_ab.py --B 32 > gpurun_out/r05_pc_ab32.txt 2>&1
for v in tmp_variants/lib_*.so; do
  t=$(basename $v .so)
  VU_LIB_PATH=$PWD/$v python tools/pcache_ab.py > gpurun_out/r05_pc_ab64_$t.txt 2>&1
done
for f in gpurun_out/r05_pc_ab*.txt; do echo "== $f"; grep -v amdgpu.ids $f | grep -v "mix_finalize\|bn_finalize\|center_dk"; done
