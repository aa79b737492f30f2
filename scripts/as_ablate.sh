#!/bin/bash
# phase ablation of the A-stationary kernel (dev library with -DUPK_DEV): usage as_ablate.sh SHAPE "cfg:ppw,cfg:ppw"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export UPK_LIB=$GRAFT_REPO_ROOT/upgpt_amd/libupk_dev.so
UPK_CXXFLAGS=-DUPK_DEV python -m upgpt_amd.build > /dev/null 2>&1
export UPK_WS_ONLY=$2
for ab in 0 0x10000 0x20000 0x40000 0x80000 0x30000 0xE0000 0xF0000 0x100000; do
  echo "--- UPK_ABLATE=$ab (10000 noepi, 20000 noBload, 40000 noLDSread, 80000 nomfma, 100000 empty)"
  UPK_ABLATE=$ab python scripts/ws_bench.py $1 2>&1 | grep "cold "
done
