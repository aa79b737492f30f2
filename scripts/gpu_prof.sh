#!/bin/bash
# rocprofv3 kernel trace (CSV stats) of the bench command. Outputs under gpurun_out/prof_*.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R; find /tmp/prof -type f | head -20
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_kernel_stats.csv; done
for f in $(find /tmp/prof -name "*kernel_trace.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Z"] if "Grid_Size_Z" in r else "", r["Workgroup_Size_X"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
with open("gpurun_out/prof_by_grid.txt", "w") as f:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        f.write("%6.2f%% n=%6d avg=%8.1fus  %s grid=%s z=%s wg=%s\n" % (100.0 * v[1] / tot, v[0], v[1] / v[0] / 1e3, k[0], k[1], k[2], k[3]))
PY
done
head -30 gpurun_out/prof_kernel_stats.csv | cut -c1-200
