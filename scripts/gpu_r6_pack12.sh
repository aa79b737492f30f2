#!/bin/bash
# Round 6: full GPU suite + a bench line with the next-weight prefetch armed for the one-batch-at-a-time configuration.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r6_pytest_gpu.txt | cut -c1-200
timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary > gpurun_out/r6_bench_pack12.json 2> gpurun_out/r6_bench_pack12.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_pack12.json").read().strip().split("\n")[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "vae_decode_ms")}, "serial", d["serial"]["value"], "fwd alone", d["unet"]["fwd_ms_graph"], "in flight", d["roofline"]["fwd_ms_per_forward_in_flight"])
PY
