#!/bin/bash
# full in-situ pass over the headline forward (B = 8, 32x32) -> gpurun_out/tuned_full.json
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
INSITU_TOPK=${1:-12} timeout 2400 python scripts/tune_insitu.py 32 32 gpurun_out/tuned_full.json 2>&1 | grep -v "^\[" | tee gpurun_out/insitu_full.log | tail -40
