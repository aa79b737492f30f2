#!/bin/bash
# phase ablation of igemm on representative shapes (graph-replayed timings)
# UPK_ABLATE bits: 0x10000 no epilogue, 0x20000 no global loads/DMA, 0x40000 no LDS writes (classic), 0x80000 no MFMA
cd $GRAFT_REPO_ROOT
for shape in "8 32 32 224 224 3 45 1" "1 8192 1 224 224 1 46 1" "1 8192 1 896 224 1 45 1" "8 8 8 896 896 3 35 8"; do
  for abl in 0 0x10000 0x20000 0x80000 0xA0000 0xB0000; do
    echo -n "abl=$abl  "; UPK_ABLATE=$abl python scripts/one_conv.py $shape 20 2>&1 | grep shape
  done
done
