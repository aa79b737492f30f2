#!/bin/bash
# Round 6, closing session on the final tree: the GPU suite, smoke(), then the evidence session (scripts/gpu_r6_evidence.sh).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest_gpu_final.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_pytest_gpu_final.txt | cut -c1-200
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
bash scripts/gpu_r6_evidence.sh
