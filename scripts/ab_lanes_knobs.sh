run() { env "$@" timeout 300 python bench.py --lanes 4 --steps 24 --warmup 4 --no-cpu-baseline --no-secondary 2>gpurun_out/b.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*:',round(d['value'],2),round(d['ms_per_step'],2),'serial',round(d['serial']['value'],2))" || tail -3 gpurun_out/b.err; }
run A=0
run UPGPT_MLP_FUSE=0 UPGPT_AUTOTUNE=1
run UPGPT_MLP_ROWS=64
run UPGPT_XBLOCK=0 UPGPT_AUTOTUNE=1
run UPGPT_HBLOCK=0 UPGPT_AUTOTUNE=1
run UPGPT_XB_ROWS=16
run A=0
