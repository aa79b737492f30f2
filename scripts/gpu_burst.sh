#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/burst scripts/ubench/burst.hip && timeout 300 /tmp/burst 2>&1 | tee gpurun_out/burst.log
