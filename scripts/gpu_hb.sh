#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_xblock_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -8
bash scripts/ab_env.sh UPGPT_HBLOCK_GN 0 1 2>&1 | tee gpurun_out/hb_ab.log
