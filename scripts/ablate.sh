#!/bin/bash
# phase ablation of igemm on representative shapes (graph-replayed timings)
cd $GRAFT_REPO_ROOT
for shape in "8 32 32 224 224 3 15 4" "8 32 32 224 224 3 17 1" "8 32 32 224 224 3 13 1" "1 8192 1 224 224 1 7 1" "8 8 8 896 896 3 1 18" "8 8 8 896 896 3 1 1"; do
  for abl in 0 0x10000 0x20000 0x40000 0x80000 0x30000 0x60000 0xF0000; do
    echo -n "abl=$abl  "; UPK_ABLATE=$abl python scripts/one_conv.py $shape 20 2>&1 | grep shape
  done
done
