#!/bin/bash
# phase ablation of the big-tile kernels on the VAE decoder's 512 -> 512 conv (UPK_ABLATE bits, dev build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f upgpt_amd/libupk.so; UPK_CXXFLAGS=-DUPK_DEV python -m upgpt_amd.build > /dev/null
for cfg in ${CFGS:-75 77 78}; do
  for abl in 0 0x10000 0x20000 0x40000 0x80000 0xC0000 0xA0000 0x60000 0xE0000; do
    echo -n "abl=$abl  "; UPK_ABLATE=$abl python scripts/one_conv.py ${SHAPE:-8 64 64 512 512 3} $cfg 1 10 2>&1 | grep shape
  done
done 2>&1 | tee gpurun_out/bt_abl.log
