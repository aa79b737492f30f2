#!/bin/bash
# One GPU-box session: tests, smoke, bench, rocprof kernel trace. Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
F='^DDIM\|Running in\|params\.\|Keeping\|Data shape\|Running DDIM\|Plotting'
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "$F" | tail -15 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -5 > gpurun_out/smoke.log; cat gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err
cd $GRAFT_REPO_ROOT; find /tmp/prof -name "*stats*" | head; for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/kernel_stats.csv; done
head -40 gpurun_out/kernel_stats.csv
