#!/bin/bash
# halo-patch family: op tests, single-shape timings, forward A/B, in-kernel timeline
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_halo_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/halo_pytest.log
{
for c in 45 81; do EPI=bias,res python scripts/one_conv.py 8 32 32 224 224 3 $c 1 20 2>&1 | grep -E "shape|Error|error" ; done
for c in 38 82 83; do EPI=bias,res python scripts/one_conv.py 8 16 16 448 448 3 $c 1 20 2>&1 | grep -E "shape|Error|error"; done
EPI=bias,res python scripts/one_conv.py 8 8 8 896 896 3 35 9 20 2>&1 | grep -E "shape|Error|error"
for sk in 2 3 4; do EPI=bias,res python scripts/one_conv.py 8 8 8 896 896 3 82 $sk 20 2>&1 | grep -E "shape|Error|error"; done
EPI=bias,res python scripts/one_conv.py 8 4 4 896 896 3 38 8 20 2>&1 | grep -E "shape|Error|error"
for sk in 4 7; do EPI=bias,res python scripts/one_conv.py 8 4 4 896 896 3 83 $sk 20 2>&1 | grep -E "shape|Error|error"; done
} | tee gpurun_out/halo_shapes.log
bash scripts/ab_env.sh UPGPT_HALO 0 8192 2048 2>&1 | tee gpurun_out/halo_ab.log
: > gpurun_out/tl_halo.log
for t in 41 55; do
  UPGPT_HALO=2048 UPK_LIB=$R/upgpt_amd/libupk_dev.so UPK_TL_TARGET=$t timeout 300 python scripts/timeline_fwd.py 2>&1 | grep -E "timeline|replay|Error" >> gpurun_out/tl_halo.log
done
cat gpurun_out/tl_halo.log
