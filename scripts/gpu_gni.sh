#!/bin/bash
# halo-patch family + input GroupNorm: tests, then the in-situ pass over the 3x3 shapes with the hc configurations only
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_halo_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/gni_pytest.log
if [ "${1:-}" = "tune" ]; then
  INSITU_ONLY3=hc INSITU_TOPK=24 timeout 1500 python scripts/tune_insitu.py 32 32 gpurun_out/tuned_gni.json 2>&1 | grep -v "^\[" | tee gpurun_out/insitu_gni.log | tail -40
fi
