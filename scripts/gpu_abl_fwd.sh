#!/bin/bash
# phase ablation of the conv / GEMM kernels INSIDE the replayed forward (dev build, results are garbage, times are not):
# UPK_ABLATE bits 0x10000 no epilogue, 0x20000 no global loads in the K loop, 0x80000 no MFMA
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export UPK_LIB=$R/upgpt_amd/libupk_dev.so
bash scripts/ab_env.sh UPK_ABLATE ${ABL_LIST:-0 0x10000 0x20000 0x80000 0x100000} 2>&1 | tee gpurun_out/abl_fwd.log
