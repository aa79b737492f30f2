export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for m in 1 2; do
  rm -rf /tmp/pf; UPK_XCD_MAP=$m rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o p -- python $R/scripts/one_conv.py 8 32 32 224 224 3 45 1 10 > /tmp/pf.log 2>&1
  f=$(find /tmp/pf -name "*counter_collection.csv" | head -1)
  python3 - "$f" $m <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "igemm" in r["Kernel_Name"]]
print("UPK_XCD_MAP=%s c3 224->224 M8192 cfg 1x7x4x1k4w3: fetch %.2f MB per launch (x2 corrected), n=%d" % (sys.argv[2], sum(float(r["Counter_Value"]) for r in rows)/len(rows)*2*1024/1e6, len(rows)))
PY
done
