#!/bin/bash
# A/B of environment knobs with the bench in its default (4-lane) configuration: one bench run per setting, same box.
run() { env "$@" timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-secondary 2>gpurun_out/b.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*:',round(d['value'],2),round(d['ms_per_step'],2),'serial',round(d['serial']['value'],2))" || tail -3 gpurun_out/b.err; }
run A=0
for kv in "$@"; do run $kv; done
run A=0
