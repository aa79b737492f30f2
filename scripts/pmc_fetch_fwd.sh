#!/bin/bash
# Fabric fetch / write per kernel of the replayed UNet forward (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE), MB per
# forward by kernel name -> gpurun_out/pmc_fetch_fwd.txt (dev tool)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pff_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pff_$c -o p -- python $R/scripts/fwd_replay.py 32 32 4 > /tmp/pff.log 2>&1 || tail -3 /tmp/pff.log
done
python - <<'PY' > $R/gpurun_out/pmc_fetch_fwd.txt
import csv, glob, collections, re
def load(c):
    f = glob.glob("/tmp/pff_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void |upkd::", "", r["Kernel_Name"]).split("(")[0][:44] + " g" + r["Grid_Size"]
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
rows = []
for k in fe:
    n = fe[k][0]
    rows.append((fe[k][1] * 2 * 1024 / 1e6 / 4, wr.get(k, [0, 0.0])[1] * 1024 / 1e6 / 4, n / 4.0, k))
rows.sort(reverse=True)
print("%10s %10s %8s  kernel (grid)   [MB per forward, 4 forwards + prep in the run]" % ("fetch", "write", "calls"))
for f, w, n, k in rows[:45]:
    print("%10.1f %10.1f %8.1f  %s" % (f, w, n, k))
print("total fetch %.0f MB write %.0f MB" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
PY
cat $R/gpurun_out/pmc_fetch_fwd.txt
