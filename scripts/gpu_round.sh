#!/bin/bash
# one GPU session: full -m gpu suite, smoke, PMC / traffic evidence, bench (driver's command), rocprof kernel stats of the
# bench, op traces (UNet + VAE), secondary configs.  Outputs under gpurun_out/ (copy what is to be judged into profiles/).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
bash scripts/gpu_r5_evidence.sh > gpurun_out/r05_evidence.log 2>&1; tail -3 gpurun_out/r05_evidence.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err; tail -c 900 gpurun_out/bench_r05.json
bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1
bash scripts/gpu_optrace.sh 32 32 > /dev/null 2>&1; tail -8 gpurun_out/ot_table_32x32.txt
bash scripts/gpu_optrace_vae.sh 32 32 > /dev/null 2>&1; tail -8 gpurun_out/ot_vae_32x32.txt
timeout 1500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cfg 3 --encoders --upscale > gpurun_out/bench_secondary.json 2> gpurun_out/bench_secondary.err; tail -c 1200 gpurun_out/bench_secondary.json
