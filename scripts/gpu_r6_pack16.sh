#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
: > gpurun_out/r6_self_prefetch_rule.txt
for v in "0 32" "2 32" "2 16" "2 8" "0 32" "2 32" "2 16"; do
  set -- $v
  UPK_SELF_PREFETCH=$1 UPK_SELF_PREFETCH_TM=$2 LAB_TAG="UPK_SELF_PREFETCH=$1 (M tiles <= $2)" timeout 300 python scripts/r6_lanes_lab.py fwd 2>/dev/null | tail -1 >> gpurun_out/r6_self_prefetch_rule.txt
done
cat gpurun_out/r6_self_prefetch_rule.txt
