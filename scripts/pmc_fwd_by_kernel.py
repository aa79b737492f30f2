"""Aggregates a rocprofv3 --pmc counter_collection.csv of scripts/fwd_replay.py per (kernel, grid): per-launch averages.
   python scripts/pmc_fwd_by_kernel.py <counter_collection.csv> [name-filter ...]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in rows:
    k = re.sub(r"\(anonymous namespace\)::|void |upkd::", "", r["Kernel_Name"]).split("(")[0][:52] + " g" + r["Grid_Size"]
    if filt and not any(f in k for f in filt):
        continue
    a = agg[k][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k in sorted(agg, key=lambda k: -max(v[0] for v in agg[k].values())):
    n = max(v[0] for v in agg[k].values())
    print("%s  (n=%d)" % (k, n))
    for c, (cnt, tot) in sorted(agg[k].items()):
        print("    %-30s per-launch %16.1f" % (c, tot / cnt))
