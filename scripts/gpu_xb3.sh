#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export UPK_LIB=$R/upgpt_amd/libupk_dev.so
for shape in "1 32" "1 256" "1 1024" "2 1024" "8 1024"; do
echo "== B hw = $shape"
python scripts/timeline_xb.py $shape 224 32 32 2>&1 | grep "trial [13]" | grep "first w[04]"
done | tee gpurun_out/xb_tl3.log
