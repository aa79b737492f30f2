#!/bin/bash
# Round 6: does the temperature of the weights matter with four chains in flight?  The same conv launches with 1 (warm), 8
# (the default of scripts/coresident.py) and 96 (> L2 + most of the Infinity Cache) rotating weight copies per lane.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
: > gpurun_out/r6_weight_temperature.txt
for shape in c3_M8192 c3_M2048 c3_M512 k1_M512; do
  for n in 1 8 96; do
    echo "## WEIGHT_COPIES=$n" >> gpurun_out/r6_weight_temperature.txt
    cfgs="4x7x2x2k2w3:1 2x7x2x2k2w3:1"
    [ $shape = c3_M2048 ] && cfgs="2x7x2x2k2w3:1 4x7x2x2k2w3:2"
    [ $shape = c3_M512 ] && cfgs="4x7x2x2k2w3:4 2x7x4x1k2w3:4"
    [ $shape = k1_M512 ] && cfgs="2x2x2x2k2w3:1 1x7x4x1k4w3:1"
    WEIGHT_COPIES=$n LAUNCHES=96 timeout 300 python scripts/coresident.py $shape $cfgs 2>/dev/null | tail -4 >> gpurun_out/r6_weight_temperature.txt
  done
done
cat gpurun_out/r6_weight_temperature.txt | cut -c1-150
