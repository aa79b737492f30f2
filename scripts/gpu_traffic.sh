#!/bin/bash
# HBM traffic of the igemm kernels per launch: separate PMC passes (FETCH_SIZE, WRITE_SIZE)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
N=4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/scripts/fwd_replay.py 32 32 $N > /tmp/pmc_$c.log 2>&1 || tail -3 /tmp/pmc_$c.log
done
python - $N <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/traffic.txt
import csv, glob, sys, json
N = int(sys.argv[1])
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    ig = [r for r in rows if "igemm" in r["Kernel_Name"]]
    main = [r for r in ig if "reduce" not in r["Kernel_Name"]]
    tot = sum(float(r["Counter_Value"]) for r in ig)
    allk = sum(float(r["Counter_Value"]) for r in rows)
    out[c] = dict(total_igemm=tot, launches=len(main), all_kernels=allk, nrows=len(rows))
print(json.dumps(out))
# prep + warm replay also contain igemm launches: main launches = 40 (prep) + (N+1 incl. capture? no) ...
PY
