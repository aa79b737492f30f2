#!/bin/bash
# A-stationary family on the GPU: parity, per-shape timing against the other configurations, phase ablation
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_astat_gpu.py -x -q > gpurun_out/astat_test.log 2>&1; tail -5 gpurun_out/astat_test.log
timeout 1500 python scripts/ws_bench.py ${SHAPES:-ff1_M8192 qkv_M8192 ff2a_M8192 to_out_M8192 ff1_M2048 qkv_M2048 ff2a_M2048 ff1_M512 qkv_M512} > gpurun_out/astat_bench.log 2>&1; grep -E "^==|as[0-9]|best|cold" gpurun_out/astat_bench.log | awk '/^==/{n=0} {if(/^==/||/best/||/as[0-9]/||n<2)print; n++}'
export UPK_LIB=$GRAFT_REPO_ROOT/upgpt_amd/libupk_dev.so
export UPK_WS_ONLY=${ABL_ONLY:-as4x2p7:4,as8x2p7:2,as4x4p7:2,2x4x2x2k2w3:1}
for ab in 0 0x10000 0x20000 0x80000 0xF0000; do
  echo "--- UPK_ABLATE=$ab (10000 noepi, 20000 noBload, 40000 noLDSread, 80000 nomfma)"
  UPK_ABLATE=$ab python scripts/ws_bench.py ff1_M8192 2>&1 | grep "cold "
done
for sk in 0 1 2 4 8; do echo "--- skew $sk"; UPK_AS_SKEW=$sk python scripts/ws_bench.py ff1_M8192 2>&1 | grep "cold "; done
