#!/bin/bash
# Round 6, third GPU session: CU-slot partitioned lanes (what the CU mask CAN do on this driver), the shader clock and the
# board power under the 4-lane load, the new GPU tests.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "upscale_config_true or sharing_one_plan" > gpurun_out/r6_pytest_new.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_pytest_new.txt
timeout 900 python scripts/r6_lanes_lab.py cumask > gpurun_out/r6_lab_cumask.txt 2> gpurun_out/r6_lab_cumask.err; echo "cumask rc $?"; tail -12 gpurun_out/r6_lab_cumask.txt | cut -c1-200; tail -3 gpurun_out/r6_lab_cumask.err
timeout 300 python scripts/r6_lanes_lab.py clock > gpurun_out/r6_lab_clock.txt 2> gpurun_out/r6_lab_clock.err; echo "clock rc $?"; cat gpurun_out/r6_lab_clock.txt; tail -3 gpurun_out/r6_lab_clock.err
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/=*//g' | tr '\n' ' '; echo; sleep 0.7; done) > gpurun_out/r6_power_under_4_lanes.txt &
SMI=$!
timeout 300 python scripts/lanes_replay.py 4 4000 > gpurun_out/r6_power_run.txt 2>&1; echo "replay rc $?"
wait $SMI; cat gpurun_out/r6_power_under_4_lanes.txt | cut -c1-200 | tail -25; tail -2 gpurun_out/r6_power_run.txt
