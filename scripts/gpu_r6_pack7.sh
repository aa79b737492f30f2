#!/bin/bash
# Round 6, seventh GPU session: tune the launches of BASELINE configs[4] at its config-true size (upscale UNet, bs 4, latent
# 3 x 128 x 96: shapes no table has yet), then the secondary bench block before / after.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python bench.py --steps 8 --warmup 4 --lanes 1 --no-cpu-baseline --no-secondary --upscale > gpurun_out/r6_upscale_before.json 2> gpurun_out/r6_upscale_before.err; echo "before rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_upscale_before.json").read().strip().split("\n")[-1])
print("before:", {k: d.get(k) for k in ("config_upscale_bs4_64x64", "config_upscale_config_true")})
PY
UPGPT_TUNE_KEEP=1 TUNE_NO_VAE=1 timeout 1500 python scripts/tune.py gpurun_out/tuned_gfx950_with_upscale_true.json upscale_true > gpurun_out/r6_tune_upscale_true.txt 2> gpurun_out/r6_tune_upscale_true.err; echo "tune rc $?"; tail -5 gpurun_out/r6_tune_upscale_true.txt
UPGPT_TUNE_FILE=$R/gpurun_out/tuned_gfx950_with_upscale_true.json timeout 300 python bench.py --steps 8 --warmup 4 --lanes 1 --no-cpu-baseline --no-secondary --upscale > gpurun_out/r6_upscale_after.json 2> gpurun_out/r6_upscale_after.err; echo "after rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_upscale_after.json").read().strip().split("\n")[-1])
print("after:", {k: d.get(k) for k in ("config_upscale_bs4_64x64", "config_upscale_config_true")})
PY
