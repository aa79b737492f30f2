#!/bin/bash
# Round 6: cooperative self-prefetch of a launch's own weight slice (UPK_SELF_PREFETCH, dev experiment): chip time per launch on
# cold weights and the three forward times, off / on.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
: > gpurun_out/r6_self_prefetch.txt
for v in 0 1 0 1; do
  UPK_SELF_PREFETCH=$v LAB_TAG="UPK_SELF_PREFETCH=$v" timeout 300 python scripts/r6_lanes_lab.py fwd 2>/dev/null | tail -1 >> gpurun_out/r6_self_prefetch.txt
done
for shape in c3_M8192 c3_M2048 c3_M512 k1_M512; do
  cfgs="4x7x2x2k2w3:1"
  [ $shape = c3_M2048 ] && cfgs="2x7x2x2k2w3:1 4x7x2x2k2w3:2"
  [ $shape = c3_M512 ] && cfgs="4x7x2x2k2w3:4"
  [ $shape = k1_M512 ] && cfgs="1x7x4x1k4w3:1"
  for v in 0 1; do
    echo "## UPK_SELF_PREFETCH=$v, 96 rotating weight copies" >> gpurun_out/r6_self_prefetch.txt
    UPK_SELF_PREFETCH=$v WEIGHT_COPIES=96 LAUNCHES=96 timeout 300 python scripts/coresident.py $shape $cfgs 2>/dev/null | tail -3 >> gpurun_out/r6_self_prefetch.txt
  done
done
cat gpurun_out/r6_self_prefetch.txt | cut -c1-160
# the driver's default bench command on this tree (wall time of the whole command)
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; echo "bench rc $? wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_default.json").read().strip().split("\n")[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "vae_decode_ms")}, "serial", d["serial"]["value"])
print("control16", d.get("control_batch16_per_forward"))
r = d["roofline"]; print({k: r.get(k) for k in ("frac", "traffic", "frac_from_trace", "l2_hit_rate")}, r.get("traffic_source", "")[:60])
print("cpu", d.get("cpu_baseline"))
PY
