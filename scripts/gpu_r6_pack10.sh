#!/bin/bash
# Round 6: next-weight prefetch (include/upk.h pf_next): forward per lane with four lanes in flight, off / on; results unchanged.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python scripts/r6_lanes_lab.py prefetch > gpurun_out/r6_lab_prefetch.txt 2> gpurun_out/r6_lab_prefetch.err; echo "prefetch rc $?"; cat gpurun_out/r6_lab_prefetch.txt | cut -c1-200; tail -3 gpurun_out/r6_lab_prefetch.err | cut -c1-300
timeout 900 python -m pytest tests/test_lanes_gpu.py tests/test_ops_gpu.py -m gpu -x -q > gpurun_out/r6_pytest_pf.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_pytest_pf.txt | cut -c1-200
