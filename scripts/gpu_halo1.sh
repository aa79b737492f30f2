#!/bin/bash
# halo-patch family: op tests, forward A/B, in-kernel timeline
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_halo_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/halo_pytest.log
bash scripts/ab_env.sh UPGPT_HALO 0 2048 128 128:400 2>&1 | tee gpurun_out/halo_ab.log
: > gpurun_out/tl_halo.log
for t in 41 55; do
  UPGPT_HALO=2048 UPK_LIB=$R/upgpt_amd/libupk_dev.so UPK_TL_TARGET=$t timeout 300 python scripts/timeline_fwd.py 2>&1 | grep -E "timeline|replay" >> gpurun_out/tl_halo.log
done
cat gpurun_out/tl_halo.log
