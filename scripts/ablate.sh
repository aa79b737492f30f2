#!/bin/bash
# phase ablation of one conv under graph replay: UPK_ABLATE bits (igemm.hip ABL_*)
cd $GRAFT_REPO_ROOT
# the hooks exist only in dev builds
rm -f upgpt_amd/libupk.so; UPK_CXXFLAGS=-DUPK_DEV python -m upgpt_amd.build > /dev/null
for cfg in ${CFGS:-45 49}; do
  for abl in 0 0x10000 0xF0000 0x100000; do
    echo -n "abl=$abl  "; UPK_ABLATE=$abl python scripts/one_conv.py ${SHAPE:-8 32 32 224 224 3} $cfg 1 20 2>&1 | grep shape
  done
done
