#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/kargs0 scripts/ubench/kargs.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -mllvm -amdgpu-kernarg-preload-count=14 -o /tmp/kargs1 scripts/ubench/kargs.hip
for i in 1 2; do echo "-- scalar loads"; timeout 120 /tmp/kargs0; echo "-- preload"; timeout 120 /tmp/kargs1; done 2>&1 | tee gpurun_out/kargs.log
