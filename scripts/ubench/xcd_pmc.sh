export TMPDIR=/tmp; cd /tmp
for sh in 0 1 8; do
  rm -rf /tmp/pm$sh
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/pm$sh -o p -- $GRAFT_REPO_ROOT/scripts/ubench/xcd_affinity $sh > /dev/null 2>&1
  f=$(find /tmp/pm$sh -name "*counter_collection.csv" | head -1)
  python3 - "$f" $sh <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items()):
    print("shift", sys.argv[2], k[0], k[1], "per launch %.0f" % (v / n), "n", n)
PY
done
