#!/bin/bash
# HIP runtime switches that touch the launch path of a replayed graph (forward per value, interleaved twice)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
bash scripts/ab_env.sh HIP_FORCE_DEV_KERNARG 1 0
bash scripts/ab_env.sh DEBUG_CLR_GRAPH_PACKET_CAPTURE 1 0
bash scripts/ab_env.sh AMD_OPT_FLUSH 1 0
bash scripts/ab_env.sh DEBUG_HIP_GRAPH_BATCH_SIZE 0 64 1024
} 2>&1 | tee gpurun_out/knobs4.log
