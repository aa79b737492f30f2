#!/bin/bash
# prices of the per-XCD engine (XCD-local barrier / hand-off, replicated weight stream), plus VERDICT r04
# item 7 (two B = 4 half batches as two branches of one graph).
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/xcdsync scripts/ubench/xcdsync.hip && timeout 300 /tmp/xcdsync 2>&1 | tee gpurun_out/xcdsync.log
timeout 600 python scripts/forkjoin.py 2>&1 | grep -v Warning | tee gpurun_out/forkjoin.log
