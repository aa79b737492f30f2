#!/bin/bash
# the same launch with random and with all-zero operands: how much of the gap to the MFMA peak is the clock under load?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for c in ${CFGS:-77 75 78 35}; do
  for z in 0 1; do echo -n "zero=$z "; ZERO=$z python scripts/one_conv.py 8 64 64 512 512 3 $c 1 200 2>&1 | grep shape; done
done
for z in 0 1; do echo -n "zero=$z "; ZERO=$z python scripts/one_conv.py 8 256 256 128 128 3 76 1 100 2>&1 | grep shape; done
for z in 0 1; do echo -n "zero=$z "; ZERO=$z python scripts/one_conv.py 8 128 128 256 256 3 78 1 100 2>&1 | grep shape; done
} | tee gpurun_out/zero.log
