#!/bin/bash
# big-tile family: correctness (every per-configuration test of test_ops_gpu) + timing on the VAE decoder's conv shapes
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv or gemm or splitk or geglu or layernorm" > gpurun_out/bt_pytest.log 2>&1; tail -3 gpurun_out/bt_pytest.log
UPK_WS_ONLY=${ONLY:-} timeout 600 python scripts/ws_bench.py ${SHAPES:-v128 v256 v512 v256_128 v128+gs+res} > gpurun_out/bt_bench.log 2>&1; grep -v "cfg [0-9]" gpurun_out/bt_bench.log
