#!/bin/bash
# Round 6, sixth GPU session: generation 2 of the throughput tiles (deep one-chunk ring, eight loader waves) by chip time;
# VAE decode with the GroupNorm statistics as conv by-product at no / every / the two highest-resolution levels.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q > gpurun_out/r6_pytest_ops.txt 2>&1; echo "pytest ops rc $?"; tail -3 gpurun_out/r6_pytest_ops.txt | cut -c1-200
: > gpurun_out/r6_gen2_chip_time.txt
run() { timeout 300 python scripts/coresident.py "$@" >> gpurun_out/r6_gen2_chip_time.txt 2>> gpurun_out/r6_gen2_chip_time.err; }
run c3_M8192 4x7x2x2k2w3:1 4x7x2x2k1w6:1 4x7x2x2k2w3l8:1 4x7x2x2k1w6l8:1 2x7x2x2k2w3:1 2x7x2x2k2w3l8:1 4x4x2x2k2w3:1 4x4x2x2k2w3l8:1
run c3_M2048 2x7x2x2k2w3:1 2x7x2x2k2w3l8:1 4x7x2x2k2w3:2 4x7x2x2k1w6:2 4x7x2x2k2w3l8:2 4x7x2x2k1w6l8:2
run c3_M512 4x7x2x2k2w3:4 4x7x2x2k1w6:4 4x7x2x2k2w3l8:4 4x7x2x2k1w6l8:4 2x7x4x1k2w3:4
run v512s 4x4x2x2k2w3:1 4x4x2x2k2w3l8:1 4x7x2x2k2w3:1 4x7x2x2k2w3l8:1
cat gpurun_out/r6_gen2_chip_time.txt | cut -c1-160; tail -5 gpurun_out/r6_gen2_chip_time.err | cut -c1-200
for v in 0 1 auto 0 auto; do UPGPT_VAE_GN_BYPRODUCT=$v timeout 300 python scripts/vae_time.py 2>/dev/null | tail -1; done | tee gpurun_out/r6_vae_gn_byproduct.txt
