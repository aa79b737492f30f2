#!/bin/bash
# per-op trace of the forward with and without the rule-based halo-patch launches
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
bash scripts/gpu_optrace.sh 32 32 > /dev/null 2>&1; cp gpurun_out/ot_table_32x32.txt gpurun_out/ot_base.txt
UPGPT_HALO=${1:-2048} bash scripts/gpu_optrace.sh 32 32 > /dev/null 2>&1; cp gpurun_out/ot_table_32x32.txt gpurun_out/ot_halo.txt
python - <<'PY'
import re
def load(p):
    rows=[]
    for l in open(p):
        m=re.match(r'\s*(\d+)\s+(\S+)\s+(.*?)\s{2,}([\d.]+)\s+(.*)$', l)
        if m: rows.append((int(m.group(1)), m.group(2), m.group(3).strip(), float(m.group(4)), m.group(5)[:60]))
    return rows
a, b = load('gpurun_out/ot_base.txt'), load('gpurun_out/ot_halo.txt')
tot=0
for x, y in zip(a, b):
    if x[1].startswith('igemm_k3') or abs(x[3]-y[3])>1.5:
        print("%3d %-9s %-42s %6.1f -> %6.1f  %+5.1f   %s | %s" % (x[0], x[1], x[2][:42], x[3], y[3], y[3]-x[3], x[4][:28], y[4][:40]))
    tot += y[3]-x[3]
print("sum of differences %.1f us" % tot)
PY
