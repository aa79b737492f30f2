#!/bin/bash
# Round-6 evidence session: everything the bench line's `roofline` block reads for the TIMED configuration
# (four forwards in flight, shared-chip table), collected for the kernel sources of THIS snapshot:
#   1. PMC passes over scripts/lanes_replay.py 4 6 (FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS | MFMA busy), control with one lane
#      on the same table  -> profiles/r06_lanes_traffic.json, r06_lanes_pmc_by_kernel.txt
#   2. rocprofv3 --kernel-trace --stats of the bench command itself -> r06_lanes_bench_by_grid.txt, r06_lanes_bench_kernel_stats.csv,
#      r06_lanes_bench_trace.json / .txt (scripts/trace_frac.py: frac_from_trace)
#   3. the bench line with those summaries in place (bench.py reads profiles/), plus the secondary block
# Outputs land in gpurun_out/ (merged back) AND in profiles/ of the box's copy so that step 3 sees them.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
COMMIT=$(cat $R/.commit 2>/dev/null || echo unknown)
SHA=$(cat $R/upgpt_amd/libupk.so.sha256)
cd /tmp
for cfg in "4 6 4" "1 24 4"; do
  set -- $cfg; tag=l$1_t$3
  rm -rf /tmp/pmc_$tag; mkdir -p /tmp/pmc_$tag
  i=0
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$tag/pass_$i -o p -- python $R/scripts/lanes_replay.py $1 $2 $3 > /tmp/pmc_$tag/pass_$i.log 2>&1 || tail -3 /tmp/pmc_$tag/pass_$i.log
    if [ "$tag" != "l4_t4" ] && [ $i -ge 2 ]; then break; fi
  done
  python $R/scripts/lanes_pmc_summary.py /tmp/pmc_$tag $1 $2 /tmp/pmc_$tag.json /tmp/pmc_$tag.txt > /dev/null 2>&1
done
LAUNCHES=$(grep -o "launches per forward: [0-9]*" /tmp/pmc_l4_t4/pass_1.log | grep -o "[0-9]*$")
python - "$COMMIT" "$SHA" "${LAUNCHES:-0}" <<'PY'
import json, sys
d = json.load(open("/tmp/pmc_l4_t4.json")); c = json.load(open("/tmp/pmc_l1_t4.json"))
d.update({"round": 6, "commit": sys.argv[1], "kernel_sources_sha256": sys.argv[2], "conv_gemm_launches_per_forward": int(sys.argv[3]) or None,
          "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE "
                     "(separate passes) -- python scripts/lanes_replay.py 4 6 4 (scripts/gpu_r6_evidence.sh)",
          "control_one_lane_same_table_fabric_bytes_per_forward": c.get("fabric_bytes_per_lane_forward"),
          "note": "rocprofv3 serialises dispatches while it collects counters (dispatch_overlap_under_pmc): the counters describe the kernels "
                  "and the launch choices of the timed configuration, each launch alone on the chip with the caches in the state the "
                  "interleaved lanes left them; GRBM_GUI_ACTIVE includes the profiler's per-dispatch overhead"})
json.dump(d, open("/tmp/r06_lanes_traffic.json", "w"), indent=1)
print(json.dumps({k: d[k] for k in ("fabric_bytes_per_lane_forward", "control_one_lane_same_table_fabric_bytes_per_forward", "l2_hit_rate", "conv_gemm_launches_per_forward") if k in d}))
PY
cp /tmp/r06_lanes_traffic.json $R/gpurun_out/r06_lanes_traffic.json; cp /tmp/r06_lanes_traffic.json $R/profiles/r06_lanes_traffic.json
cp /tmp/pmc_l4_t4.txt $R/gpurun_out/r06_lanes_pmc_by_kernel.txt
# 2. kernel trace of the bench command
rm -rf /tmp/profl; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profl -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $R/gpurun_out/r06_lanes_bench_under_rocprof.json 2> $R/gpurun_out/r06_prof.err
cd $R
for f in $(find /tmp/profl -name "*kernel_stats.csv"); do cp $f gpurun_out/r06_lanes_bench_kernel_stats.csv; done
for f in $(find /tmp/profl -name "*kernel_trace.csv"); do
  python scripts/trace_frac.py "$f" /tmp/r06_trace.json gpurun_out/r06_lanes_bench_trace_windows.txt > /dev/null
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:60], r["Grid_Size_X"], r.get("Grid_Size_Z", ""), r["Workgroup_Size_X"])
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
with open("gpurun_out/r06_lanes_bench_by_grid.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace of `python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary` (4 lanes): share of summed kernel time, launches, avg duration\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        f.write("%6.2f%% n=%6d avg=%8.1fus  %s grid=%s z=%s wg=%s\n" % (100.0 * v[1] / tot, v[0], v[1] / v[0] / 1e3, k[0], k[1], k[2], k[3]))
PY
done
python - "$COMMIT" "$SHA" <<'PY'
import json, sys
d = json.load(open("/tmp/r06_trace.json"))
d.update({"round": 6, "commit": sys.argv[1], "kernel_sources_sha256": sys.argv[2],
          "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary; scripts/trace_frac.py over the kernel trace (scripts/gpu_r6_evidence.sh)"})
json.dump(d, open("/tmp/r06_lanes_bench_trace.json", "w"), indent=1)
print(json.dumps({k: d.get(k) for k in ("frac_from_trace", "union_ms_per_forward", "kernels_running_while_busy", "forwards", "steady_windows")}))
PY
cp /tmp/r06_lanes_bench_trace.json gpurun_out/r06_lanes_bench_trace.json; cp /tmp/r06_lanes_bench_trace.json profiles/r06_lanes_bench_trace.json
head -8 gpurun_out/r06_lanes_bench_by_grid.txt | cut -c1-170
# 3. the bench line (default arguments of the driver) + secondary block
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc $?"
timeout 900 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --cfg 3.0 --encoders --upscale > gpurun_out/r06_bench_secondary.json 2> gpurun_out/r06_bench_secondary.err; echo "secondary rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench.json").read().strip().split("\n")[-1])
r = d["roofline"]
print({k: d.get(k) for k in ("value", "ms_per_step", "vae_decode_ms", "mfma_util_whole_job")}, d.get("serial", {}).get("value"))
print({k: r.get(k) for k in ("frac", "traffic", "frac_from_trace", "fabric_bytes_per_lane_forward", "l2_hit_rate", "fwd_ms_per_forward_in_flight", "traffic_source", "frac_from_trace_source")})
PY
