#!/bin/bash
# Round 6, fourth GPU session: more than four lanes on full-mask streams (own hardware queues), row-chain kernels at the 16x16
# level with the chip shared.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for m in morelanes knobs; do
  timeout 900 python scripts/r6_lanes_lab.py $m > gpurun_out/r6_lab_$m.txt 2> gpurun_out/r6_lab_$m.err; echo "lab $m rc $?"
  cat gpurun_out/r6_lab_$m.txt | cut -c1-250; tail -3 gpurun_out/r6_lab_$m.err
done
