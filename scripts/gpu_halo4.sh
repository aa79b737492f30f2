#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_halo_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/halo_pytest.log
{
for c in 45 81; do EPI=bias,res python scripts/one_conv.py 8 32 32 224 224 3 $c 1 20 2>&1 | grep -E "shape|rror" ; done
for c in 45 81; do EPI=bias,res python scripts/one_conv.py 8 32 32 448 224 3 $c 1 20 2>&1 | grep -E "shape|rror" ; done
} | tee gpurun_out/halo_shapes.log
bash scripts/ab_env.sh UPGPT_HALO 0 8192 2>&1 | tee gpurun_out/halo_ab.log
: > gpurun_out/tl_halo.log
for t in 41 42; do
  UPGPT_HALO=8192 UPK_LIB=$R/upgpt_amd/libupk_dev.so UPK_TL_TARGET=$t timeout 300 python scripts/timeline_fwd.py 2>&1 | grep -E "timeline|replay|Error" >> gpurun_out/tl_halo.log
done
cat gpurun_out/tl_halo.log
