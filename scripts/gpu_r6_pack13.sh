#!/bin/bash
# same-box A/B of the next-weight prefetch on the one-batch-at-a-time bench (--lanes 1)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 auto 0 auto; do
  UPGPT_WEIGHT_PREFETCH=$v timeout 300 python bench.py --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('UPGPT_WEIGHT_PREFETCH=$v: %.2f images/s, %.2f ms per step, UNet forward %.3f ms, decode %.2f ms' % (d['value'], d['ms_per_step'], d['unet']['fwd_ms_graph'], d['vae_decode_ms']))"
done | tee gpurun_out/r6_prefetch_serial_bench_ab.txt
