#!/bin/bash
# Round 6, fifth GPU session: the 224 / 112-column big-tile configurations (bt4x7*): op tests, then chip time per launch with
# four chains of the same conv in flight against the shared-chip table's choices (scripts/coresident.py).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_bigtile_gpu.py -m gpu -x -q > gpurun_out/r6_pytest_bt.txt 2>&1; echo "pytest bt rc $?"; tail -3 gpurun_out/r6_pytest_bt.txt | cut -c1-200
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "upscale_config_true or sharing_one_plan" > gpurun_out/r6_pytest_new.txt 2>&1; echo "pytest new rc $?"; tail -3 gpurun_out/r6_pytest_new.txt | cut -c1-200
: > gpurun_out/r6_bt7_chip_time.txt
run() { timeout 300 python scripts/coresident.py "$@" >> gpurun_out/r6_bt7_chip_time.txt 2>> gpurun_out/r6_bt7_chip_time.err; }
run c3_M8192 4x7x2x2k2w3:1 bt4x7x2x2n3:1 bt4x7x4x1n3:1 bt4x7x4x2n4:1 bt4x7x4x2n4:2 bt4x7x2x2n3:2 2x7x2x2k2w3:1
run c3_M2048 2x7x2x2k2w3:1 4x7x2x2k2w3:1 4x7x2x2k2w3:2 bt4x7x2x2n3:1 bt4x7x2x2n3:2 bt4x7x4x1n3:1 bt4x7x4x1n3:2 bt4x7x4x2n4:2 bt4x7x4x2n4:4
run c3_M512 4x2x2x2k2w3:4 2x7x4x1k2w3:4 4x7x2x2k2w3:4 bt4x7x2x2n3:4 bt4x7x2x2n3:8 bt4x7x4x1n3:4 bt4x7x4x1n3:8 bt4x7x4x2n4:8
run k1_M2048 2x2x2x2k2w3:1 2x4x2x2k2w3:1 bt4x7x2x2n3:1 bt4x7x4x1n3:1
cat gpurun_out/r6_bt7_chip_time.txt | cut -c1-160; tail -5 gpurun_out/r6_bt7_chip_time.err | cut -c1-200
