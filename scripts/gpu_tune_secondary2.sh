#!/bin/bash
# in-situ passes over the other secondary forwards with the current kernels: guidance (B = 16 rows) and the true 256x192 size
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
INSITU_B=16 INSITU_TOPK=10 timeout 1500 python scripts/tune_insitu.py 32 32 gpurun_out/tuned_cfg.json 2>&1 | grep -v "^\[" | tee gpurun_out/insitu_cfg.log | tail -25
UPGPT_TUNE_FILE=$R/gpurun_out/tuned_cfg.json INSITU_TOPK=10 timeout 1500 python scripts/tune_insitu.py 32 24 gpurun_out/tuned_cfg_3224.json 2>&1 | grep -v "^\[" | tee gpurun_out/insitu_3224.log | tail -25
