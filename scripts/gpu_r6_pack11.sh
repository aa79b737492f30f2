#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python scripts/r6_lanes_lab.py prefetch > gpurun_out/r6_lab_prefetch2.txt 2> gpurun_out/r6_lab_prefetch2.err; echo "prefetch rc $?"; cat gpurun_out/r6_lab_prefetch2.txt | cut -c1-220; tail -3 gpurun_out/r6_lab_prefetch2.err | cut -c1-300
