#!/bin/bash
# xblock: tests, then A/B in the forward
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_xblock_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/xb_test.log
cat gpurun_out/xb_test.log
bash scripts/ab_env.sh UPGPT_XBLOCK 0 auto 1 2>&1 | tee gpurun_out/xb_ab.log
UPGPT_XBLOCK=1 bash scripts/ab_env.sh UPGPT_XB_ROWS 16 32 2>&1 | tee -a gpurun_out/xb_ab.log
