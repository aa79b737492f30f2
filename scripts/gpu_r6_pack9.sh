#!/bin/bash
# Round 6: cold weights (96 rotating copies) private to each lane vs SHARED by the four lanes, which launch the same sequence
# at the same time — the upper bound of what keeping the lanes of one model aligned layer by layer could give.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
: > gpurun_out/r6_weight_sharing.txt
for shape in c3_M8192 c3_M2048 c3_M512 k1_M512 k1_M2048; do
  for sh in 0 1; do
    echo "## 96 rotating weight copies, SHARE_WEIGHTS=$sh" >> gpurun_out/r6_weight_sharing.txt
    cfgs="4x7x2x2k2w3:1"
    [ $shape = c3_M2048 ] && cfgs="2x7x2x2k2w3:1"
    [ $shape = c3_M512 ] && cfgs="4x7x2x2k2w3:4"
    [ $shape = k1_M512 ] && cfgs="1x7x4x1k4w3:1"
    [ $shape = k1_M2048 ] && cfgs="2x2x2x2k2w3:1"
    SHARE_WEIGHTS=$sh WEIGHT_COPIES=96 LAUNCHES=96 timeout 300 python scripts/coresident.py $shape $cfgs 2>/dev/null | tail -3 >> gpurun_out/r6_weight_sharing.txt
  done
done
cat gpurun_out/r6_weight_sharing.txt | cut -c1-150
