#!/bin/bash
# A/B of builds of the library on one box: forward time under graph replay, interleaved twice
#   bash scripts/gpu_ab_lib.sh upgpt_amd/libupk.so upgpt_amd/libupk_nt.so ...
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
libs=""; for l in "$@"; do libs="$libs $R/$l"; done
bash scripts/ab_env.sh UPK_LIB $libs 2>&1 | sed "s#$R/##" | tee gpurun_out/ab_lib.log
