#!/bin/bash
# in-situ tuning of the shapes whose key contains $1 (headline forward, B = 8, 32x32) -> gpurun_out/tuned_keys.json
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
INSITU_KEYS="$1" INSITU_TOPK=${2:-24} timeout 1500 python scripts/tune_insitu.py 32 32 gpurun_out/tuned_keys.json 2>&1 | grep -v "^\[" | tee gpurun_out/insitu_keys.log | tail -30
