#!/bin/bash
# closing check on the final tree: GPU suite, smoke, the driver's bench command, the secondary block
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest_gpu_final.txt 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r6_pytest_gpu_final.txt | cut -c1-200
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc $? wall $(( $(date +%s) - S )) s"
timeout 900 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --cfg 3.0 --encoders --upscale > gpurun_out/r06_bench_secondary.json 2> gpurun_out/r06_bench_secondary.err; echo "secondary rc $?"; tail -2 gpurun_out/r06_bench_secondary.err | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench.json").read().strip().split("\n")[-1])
r = d["roofline"]
print({k: d.get(k) for k in ("value", "ms_per_step", "vae_decode_ms", "mfma_util_whole_job")}, "serial", d["serial"]["value"], "control16", d.get("control_batch16_per_forward", {}).get("value"))
print({k: r.get(k) for k in ("frac", "traffic", "frac_from_trace", "l2_hit_rate", "fwd_ms_per_forward_in_flight")}, str(r.get("traffic_source"))[:40])
s = json.loads(open("gpurun_out/r06_bench_secondary.json").read().strip().split("\n")[-1])
print({k: (s[k].get("value") if isinstance(s.get(k), dict) else None) for k in ("config_cfg", "config_full_cond_with_encoders", "config_upscale_bs4_64x64", "config_upscale_config_true")})
PY
