#!/bin/bash
# re-check of the launch-geometry knobs on the current kernels (forward time per value, interleaved twice each)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
bash scripts/ab_env.sh UPK_ATTN_WPB 0 1 2 4
bash scripts/ab_env.sh UPK_GN_BLOCK_ELEMS 2048 1024 4096
bash scripts/ab_env.sh UPK_GNAPPLY_NVMAX 2 1 4
bash scripts/ab_env.sh UPGPT_XB_ROWS 0 16
bash scripts/ab_env.sh UPGPT_QPROJ_FUSE 1 0
bash scripts/ab_env.sh UPGPT_HBLOCK_GN 1 0
bash scripts/ab_env.sh UPGPT_GN_REDUCE_APPLY 1 0
} 2>&1 | tee gpurun_out/knobs.log
