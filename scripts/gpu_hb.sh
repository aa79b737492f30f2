#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_xblock_gpu.py -x -q 2>&1 | tail -15
bash scripts/ab_env.sh UPGPT_HBLOCK 0 auto 2>&1 | tee gpurun_out/hb_ab.log
bash scripts/gpu_optrace.sh 32 32 > /dev/null 2>&1
grep -E "hblock|xblock|mlp " gpurun_out/ot_table_32x32.txt | head; tail -6 gpurun_out/ot_table_32x32.txt
