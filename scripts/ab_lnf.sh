#!/bin/bash
# A/B on one box: previous library (save it as scripts/ab/libupk_old.so before rebuilding; git-ignored) vs the current build
mkdir -p gpurun_out; export TMPDIR=/tmp
fwd() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], 'img/s %.2f fwd_ms %.4f' % (d['value'], d['unet']['fwd_ms_graph']), d['unet']['class_ms_per_fwd'])" $1; }
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "layernorm or all_configs or appended" 2>&1 | tail -2
cp upgpt_amd/libupk.so /tmp/libupk_new.so
for r in 1 2; do
cp scripts/ab/libupk_old.so upgpt_amd/libupk.so
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/lnf_old$r.json 2>/dev/null; fwd gpurun_out/lnf_old$r.json
cp /tmp/libupk_new.so upgpt_amd/libupk.so
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/lnf_new$r.json 2>/dev/null; fwd gpurun_out/lnf_new$r.json
done
