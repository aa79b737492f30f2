#!/bin/bash
# Fabric traffic (FETCH_SIZE, x2 gfx950 correction) of single conv/GEMM shapes under graph replay (dev tool).
#   bash scripts/pmc_fetch.sh  -> gpurun_out/pmc_fetch.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
: > $R/gpurun_out/pmc_fetch.txt
run() { # name alg_MB args...
  name=$1; alg=$2; shift; shift
  rm -rf /tmp/pf; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o p -- python $R/scripts/one_conv.py "$@" > /tmp/pf.log 2>&1 || tail -3 /tmp/pf.log
  f=$(find /tmp/pf -name "*counter_collection.csv" | head -1)
  python - "$f" "$name" "$alg" >> $R/gpurun_out/pmc_fetch.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if "igemm" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0][-40:]
    a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, v in agg.items():
    print("%-14s %-42s fetch %8.2f MB per launch (x2 corrected), algorithmic %s MB, n=%d" % (sys.argv[2], k, v[1] / v[0] * 2 * 1024 / 1e6, sys.argv[3], v[0]))
PY
  grep shape /tmp/pf.log >> $R/gpurun_out/pmc_fetch.txt
}
# B H W cin cout ks cfg sk reps  (cfg indices: see upk_conv_config_name)
run c3_M512_sk9 15.4 8 8 8 896 896 3 35 9 10
run ff2_M512_sk8 7.3 1 512 1 3584 896 1 59 8 10
run k1_M512 2.5 1 512 1 896 896 1 53 1 10
run c3_M128_sk9 14.7 8 4 4 896 896 3 38 9 10
EPI=geglu run ff1_M512 13.8 1 512 1 896 7168 1 57 1 10
run qkv_M512 6.4 1 512 1 896 3072 1 47 1 10
run c3_M2048 7.3 8 16 16 448 448 3 38 1 10
run ff1_M2048in 5.0 1 2048 1 448 3584 1 36 1 10
cat $R/gpurun_out/pmc_fetch.txt
