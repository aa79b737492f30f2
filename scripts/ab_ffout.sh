#!/bin/bash
# tune the fused ff.net.2+proj_out shapes, then bench: fold off / auto (tuned) on the same box
mkdir -p gpurun_out; export TMPDIR=/tmp
F='^DDIM\|Running in\|params\.\|Keeping\|Data shape\|Running DDIM\|Plotting'
fwd() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], 'img/s %.2f fwd_ms %.4f launches %d' % (d['value'], d['unet']['fwd_ms_graph'], d['unet']['kernel_launches_per_fwd']), d['unet']['class_ms_per_fwd'])" $1; }
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -s -k "skip_projection" 2>&1 | grep -i "mse\|passed\|failed\|error\|assert" | tail -8
UPGPT_TUNE_KEEP=1 timeout 1500 python scripts/tune.py gpurun_out/tuned_ff.json bbox,bbox_cfg,upscale 2>&1 | grep -v "$F" | grep "skip_fold=1\|entries" | tail -12
UPGPT_FFOUT_FOLD=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ff_off.json 2>/dev/null; fwd gpurun_out/ff_off.json
UPGPT_TUNE_FILE=gpurun_out/tuned_ff.json timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ff_auto.json 2>/dev/null; fwd gpurun_out/ff_auto.json
UPGPT_FFOUT_FOLD=1 UPGPT_TUNE_FILE=gpurun_out/tuned_ff.json timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ff_all.json 2>/dev/null; fwd gpurun_out/ff_all.json
python - <<'PY'
import json
t = json.load(open("gpurun_out/tuned_ff.json")); a = json.load(open("upgpt_amd/tuned_gfx950.json"))
for k, v in sorted(t.items()):
    if k not in a:
        print(k, v)
PY
