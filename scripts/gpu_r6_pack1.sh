#!/bin/bash
# Round 6, first GPU session: the GPU suite after the housekeeping refactor (per-XCD engine removed, scoped concurrency),
# then the lane experiments of VERDICT r05 items 3, 4, 6 (scripts/r6_lanes_lab.py) and a bench line.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_pytest_gpu.txt
tail -3 gpurun_out/r6_pytest_gpu.txt
for m in cumask align controls; do
  timeout 900 python scripts/r6_lanes_lab.py $m > gpurun_out/r6_lab_$m.txt 2> gpurun_out/r6_lab_$m.err; echo "lab $m rc $?"
  tail -14 gpurun_out/r6_lab_$m.txt; tail -3 gpurun_out/r6_lab_$m.err
done
timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary > gpurun_out/r6_bench_pack1.json 2> gpurun_out/r6_bench_pack1.err; echo "bench rc $?"
tail -c 1500 gpurun_out/r6_bench_pack1.json
