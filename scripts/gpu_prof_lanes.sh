#!/bin/bash
# rocprofv3 kernel trace of the bench command in its default configuration (4 execution lanes): per-kernel stats, the trace
# by (kernel, grid), and how much the kernels of different lanes overlap (sum of kernel durations vs the union of their
# intervals, inside the timed region's span).  Outputs under gpurun_out/prof_lanes_*.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/profl && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profl -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_lanes_bench.json 2> $R/gpurun_out/prof_lanes.err
cd $R
for f in $(find /tmp/profl -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_lanes_kernel_stats.csv; done
for f in $(find /tmp/profl -name "*kernel_trace.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
iv = []
for r in rows:
    k = (r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Z"] if "Grid_Size_Z" in r else "", r["Workgroup_Size_X"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
    iv.append((s, e, r.get("Queue_Id", "?")))
tot = sum(v[1] for v in agg.values())
with open("gpurun_out/prof_lanes_by_grid.txt", "w") as f:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        f.write("%6.2f%% n=%6d avg=%8.1fus  %s grid=%s z=%s wg=%s\n" % (100.0 * v[1] / tot, v[0], v[1] / v[0] / 1e3, k[0], k[1], k[2], k[3]))
# overlap: windows of 100 ms over the whole run; in each, sum of kernel time / union of kernel intervals, kernels per queue
iv.sort()
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
W = 100_000_000
with open("gpurun_out/prof_lanes_overlap.txt", "w") as f:
    f.write("# rocprofv3 kernel trace of `python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary` (4 lanes): per 100-ms window\n")
    f.write("# window_start_ms  kernels  queues  sum_of_kernel_durations_ms  union_busy_ms  avg_kernels_running_while_busy  busy_share_of_window\n")
    w = t0
    while w < t1:
        sel = [(max(s, w), min(e, w + W), q) for s, e, q in iv if e > w and s < w + W]
        if sel:
            ssum = sum(e - s for s, e, _ in sel)
            union, cur_s, cur_e = 0, None, None
            for s, e, _ in sorted(sel):
                if cur_e is None or s > cur_e:
                    if cur_e is not None: union += cur_e - cur_s
                    cur_s, cur_e = s, e
                else:
                    cur_e = max(cur_e, e)
            union += cur_e - cur_s
            f.write("%10.1f %8d %6d %12.2f %12.2f %8.2f %8.3f\n" % ((w - t0) / 1e6, len(sel), len(set(q for _, _, q in sel)), ssum / 1e6, union / 1e6, ssum / max(1, union), union / W))
        w += W
PY
done
head -12 gpurun_out/prof_lanes_by_grid.txt | cut -c1-160; tail -25 gpurun_out/prof_lanes_overlap.txt; tail -c 600 gpurun_out/prof_lanes_bench.json
