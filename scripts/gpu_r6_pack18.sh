#!/bin/bash
# needs the experiments library built from commit e4f4ad5 (the staggered walk is not in the tree any more):
#   UPK_LIB=$PWD/upgpt_amd/libupk_exp.so UPK_CXXFLAGS=-DUPK_R6_EXPERIMENTS python -m upgpt_amd.build
# Round 6: staggered K walk (UPK_KROT, -DUPK_R6_EXPERIMENTS build): op tests both ways, chip time per launch on cold weights, the
# three forward times.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export UPK_LIB=$R/upgpt_amd/libupk_exp.so
for v in 0 1; do UPK_KROT=$v timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q > gpurun_out/r6_pytest_krot$v.txt 2>&1; echo "ops tests UPK_KROT=$v rc $?"; tail -3 gpurun_out/r6_pytest_krot$v.txt | cut -c1-200; done
: > gpurun_out/r6_krot.txt
for v in 0 1 0 1; do
  UPK_KROT=$v LAB_TAG="UPK_KROT=$v" timeout 300 python scripts/r6_lanes_lab.py fwd 2>/dev/null | tail -1 >> gpurun_out/r6_krot.txt
done
for shape in c3_M8192 c3_M2048 c3_M512 k1_M2048; do
  cfgs="4x7x2x2k2w3:1"
  [ $shape = c3_M2048 ] && cfgs="2x7x2x2k2w3:1 4x7x2x2k2w3:2"
  [ $shape = c3_M512 ] && cfgs="4x7x2x2k2w3:4"
  [ $shape = k1_M2048 ] && cfgs="2x2x2x2k2w3:1"
  for v in 0 1; do
    echo "## UPK_KROT=$v, 96 rotating weight copies" >> gpurun_out/r6_krot.txt
    UPK_KROT=$v WEIGHT_COPIES=96 LAUNCHES=96 timeout 300 python scripts/coresident.py $shape $cfgs 2>/dev/null | tail -3 >> gpurun_out/r6_krot.txt
  done
done
cat gpurun_out/r6_krot.txt | cut -c1-160
