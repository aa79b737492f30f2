#!/bin/bash
# needs the dev library: UPK_LIB=$PWD/upgpt_amd/libupk_dev.so UPK_CXXFLAGS=-DUPK_DEV python -m upgpt_amd.build (built here, ships with the snapshot)
# Round 6: what do the operand fetches cost with four chains in flight?  Dev build (-DUPK_DEV): the wave-specialised loaders fetch
# the zero page instead of the A (im2col) rows / the B (weight) rows / both — the DMA instructions stay, their lines do not
# (results are garbage, times are not).  Upper bound of what a halo-resident A operand (A bytes / 5.6) could give.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export UPK_LIB=$R/upgpt_amd/libupk_dev.so
: > gpurun_out/r6_operand_ablation.txt
for shape in c3_M8192 c3_M2048 c3_M512 k1_M2048; do
  cfgs="4x7x2x2k2w3:1"
  [ $shape = c3_M2048 ] && cfgs="2x7x2x2k2w3:1"
  [ $shape = c3_M512 ] && cfgs="4x7x2x2k2w3:4"
  [ $shape = k1_M2048 ] && cfgs="2x2x2x2k2w3:1"
  for abl in 0 0x4000000 0x8000000 0xC000000 0x20000 0x80000; do
    echo "## UPK_ABLATE=$abl (0x4000000 no A lines, 0x8000000 no B lines, 0xC000000 neither, 0x20000 no DMAs at all, 0x80000 no MFMAs)" >> gpurun_out/r6_operand_ablation.txt
    UPK_ABLATE=$abl WEIGHT_COPIES=96 LAUNCHES=96 timeout 300 python scripts/coresident.py $shape $cfgs 2>/dev/null | tail -2 >> gpurun_out/r6_operand_ablation.txt
  done
done
cat gpurun_out/r6_operand_ablation.txt | cut -c1-170
