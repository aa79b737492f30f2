#!/bin/bash
# phase ablation of the patch kernel on one shape / configuration (dev library with -DUPK_DEV)
# usage: pc_ablate.sh SHAPE CFGS   (e.g. L0a 10,4)
export UPK_LIB=$GRAFT_REPO_ROOT/upgpt_amd/libupk_dev.so
S=$1; export UPK_PC_CFGS=$2
for ab in 0 0x10000 0x20000 0x40000 0x80000 0x100000 0xF0000; do
  echo "--- UPK_ABLATE=$ab (10000 noepi, 20000 noBdma, 40000 nostage, 80000 nomfma, 100000 no-transform)"
  UPK_ABLATE=$ab python scripts/pc_bench.py $S 2>&1 | grep " us "
done
