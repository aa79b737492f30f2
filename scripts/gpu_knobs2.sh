#!/bin/bash
# engine-level fold decisions re-checked in situ (they are taken from single-launch tuning-table times by default)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
bash scripts/ab_env.sh UPGPT_SKIP_FOLD auto 1 0
bash scripts/ab_env.sh UPGPT_FFOUT_FOLD auto 1 0
bash scripts/ab_env.sh UPGPT_LN_FOLD auto 1 0
bash scripts/ab_env.sh UPGPT_MLP_ROWS 0 64
} 2>&1 | tee gpurun_out/knobs3.log
