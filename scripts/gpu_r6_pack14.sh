#!/bin/bash
# Round 6: the driver's sequence on the final tree — GPU suite, smoke, default bench (wall time of the whole command).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest_gpu_final.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_pytest_gpu_final.txt | cut -c1-200
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
/usr/bin/time -v timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; echo "bench rc $?"; grep "Elapsed (wall" gpurun_out/r6_bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_default.json").read().strip().split("\n")[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "vae_decode_ms")}, "serial", d["serial"]["value"])
print("control16", d.get("control_batch16_per_forward"))
print("cfg_true", {k: v for k, v in d.get("config_true_256x192", {}).items() if k != "serial"})
r = d["roofline"]; print({k: r.get(k) for k in ("frac", "traffic", "frac_from_trace", "l2_hit_rate", "mfma_busy_over_gui_active")})
print("cpu", d.get("cpu_baseline"))
PY
