#!/bin/bash
# GPU clock under load from counters: GRBM_GUI_ACTIVE (GPU-clock cycles the kernel was active) / kernel duration, for
# the same big-tile launch with random and with all-zero operands; plus the MFMA-busy share of those cycles.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
for z in 0 1; do for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rm -rf /tmp/cp_$z_$c; ZERO=$z rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cp_${z}_$c -o p -- python $R/scripts/one_conv.py 8 64 64 512 512 3 ${CFG:-75} 1 60 > /tmp/cp_${z}_$c.log 2>&1 || tail -3 /tmp/cp_${z}_$c.log
done; done
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/clkpmc.txt
import csv, glob
for z in (0, 1):
    dur = None; vals = {}
    for c in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"):
        fs = glob.glob("/tmp/cp_%d_%s/**/*counter_collection.csv" % (z, c), recursive=True)
        if not fs: continue
        rows = [r for r in csv.DictReader(open(fs[0])) if "igemm_bt" in r["Kernel_Name"] and r["Counter_Name"] == c]
        if rows: vals[c] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
        ts = glob.glob("/tmp/cp_%d_%s/**/*kernel_trace.csv" % (z, c), recursive=True)
        if ts:
            k = [r for r in csv.DictReader(open(ts[0])) if "igemm_bt" in r["Kernel_Name"]]
            if k: dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in k) / len(k) / 1e3
    print("operands %s: kernel %.1f us under the counters; per launch: %s" % ("zero" if z else "random", dur or -1, vals))
    if dur and "GRBM_GUI_ACTIVE" in vals:
        print("   GRBM_GUI_ACTIVE / duration = %.0f MHz" % (vals["GRBM_GUI_ACTIVE"] / dur))
PY
