#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
INSITU_ONLY3=hc INSITU_TOPK=40 timeout 1500 python scripts/tune_insitu.py 32 32 gpurun_out/tuned_hc.json 2>&1 | grep -v "^\[" | tee gpurun_out/insitu_hc.log | tail -40
