#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_xblock_gpu.py -x -q 2>&1 | tail -3
bash scripts/gpu_optrace.sh 32 32 > /dev/null 2>&1; cp gpurun_out/ot_table_32x32.txt gpurun_out/ot_touch0.txt
UPGPT_TOUCH=each bash scripts/gpu_optrace.sh 32 32 > /dev/null 2>&1; cp gpurun_out/ot_table_32x32.txt gpurun_out/ot_touch1.txt
tail -12 gpurun_out/ot_touch0.txt; tail -12 gpurun_out/ot_touch1.txt
bash scripts/ab_env.sh UPGPT_TOUCH 0 each
