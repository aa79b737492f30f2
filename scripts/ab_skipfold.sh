#!/bin/bash
# A/B of the appended-K-segment change: old library vs new library (fold off) vs new library + tuned fold.
# (needs the PREVIOUS build of the library saved as scripts/ab/libupk_old.so before rebuilding; git-ignored)
mkdir -p gpurun_out; export TMPDIR=/tmp
F='^DDIM\|Running in\|params\.\|Keeping\|Data shape\|Running DDIM\|Plotting'
fwd() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], 'img/s %.2f fwd_ms %.4f' % (d['value'], d['unet']['fwd_ms_graph']), d['unet']['class_ms_per_fwd'])" $1; }
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "appended or all_configs or randomised_shapes" 2>&1 | tail -3
cp upgpt_amd/libupk.so /tmp/libupk_new.so
cp scripts/ab/libupk_old.so upgpt_amd/libupk.so
UPGPT_SKIP_FOLD=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ab_old.json 2>/dev/null; fwd gpurun_out/ab_old.json
cp /tmp/libupk_new.so upgpt_amd/libupk.so
UPGPT_SKIP_FOLD=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ab_new_nofold.json 2>/dev/null; fwd gpurun_out/ab_new_nofold.json
UPGPT_TUNE_KEEP=1 timeout 1500 python scripts/tune.py gpurun_out/tuned_ka.json bbox 2>&1 | grep -v "$F" | tail -12
UPGPT_TUNE_FILE=gpurun_out/tuned_ka.json timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ab_new_fold.json 2>/dev/null; fwd gpurun_out/ab_new_fold.json
UPGPT_SKIP_FOLD=1 UPGPT_TUNE_FILE=gpurun_out/tuned_ka.json timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ab_new_foldall.json 2>/dev/null; fwd gpurun_out/ab_new_foldall.json
python - <<'PY'
import json
t = json.load(open("gpurun_out/tuned_ka.json"))
for k, v in sorted(t.items()):
    if "_ka" in k:
        print(k, v)
PY
