#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_xblock_gpu.py -x -q 2>&1 | tail -5
bash scripts/ab_env.sh UPGPT_XBLOCK 0 1 2>&1 | tee gpurun_out/xb_ab.log
export UPGPT_XBLOCK=1
bash scripts/gpu_optrace.sh 32 32 > /dev/null 2>&1
grep -E "xblock" gpurun_out/ot_table_32x32.txt | head -40
export UPK_LIB=$R/upgpt_amd/libupk_dev.so
python scripts/timeline_xb.py 8 1024 224 32 32 2>&1 | grep "trial [13]" | tee gpurun_out/xb_tl.log
python scripts/timeline_xb.py 8 256 448 64 16 2>&1 | grep "trial [13]" | tee -a gpurun_out/xb_tl.log
