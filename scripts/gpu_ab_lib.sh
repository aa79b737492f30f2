#!/bin/bash
# A/B of two builds of the library on one box: forward time under graph replay, interleaved twice
#   bash scripts/gpu_ab_lib.sh upgpt_amd/libupk_base.so upgpt_amd/libupk.so
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
bash scripts/ab_env.sh UPK_LIB "$R/$1" "$R/$2" 2>&1 | sed "s#$R/##" | tee gpurun_out/ab_lib.log
