cd $GRAFT_REPO_ROOT
bash tools/gpu_trace.sh r06o > gpurun_out/r06o_trace.log 2>&1
grep -i "attn_f1\|scores\|mix_center\|map_rows\|map_cols\|map_bwd\|bn_\|total us" gpurun_out/r06o_by_grid.csv gpurun_out/r06o_trace.log | cut -c1-230 | head -30
