#!/bin/bash
# needs upgpt_amd/libupk_old.so = the library of commit b450d6c's csrc/ (UPK_LIB=... python -m upgpt_amd.build on that checkout)
# same-box A/B: library before / after the loader's seek() refactor (commit b450d6c sources vs the tree)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
: > gpurun_out/r6_ab_seek_refactor.txt
for v in old new old new; do
  if [ $v = old ]; then export UPK_LIB=$R/upgpt_amd/libupk_old.so; else unset UPK_LIB; fi
  LAB_TAG="library $v" timeout 300 python scripts/r6_lanes_lab.py fwd 2>/dev/null | tail -1 >> gpurun_out/r6_ab_seek_refactor.txt
done
cat gpurun_out/r6_ab_seek_refactor.txt
