#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in 48 49 54; do
  for abl in 0 0x10000 0x20000 0x80000 0xB0000 0xF0000; do
    echo -n "abl=$abl  "; UPK_ABLATE=$abl python scripts/one_conv.py 8 32 32 224 224 3 $cfg 1 20 2>&1 | grep shape
  done
done
