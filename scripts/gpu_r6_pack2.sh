#!/bin/bash
# Round 6, second GPU session: what a CU mask does on this box (maskdiag), the (lanes, B) controls, clocks / power under
# the 4-lane load, and the counters of the TIMED configuration: FETCH / WRITE / TCC / MFMA-busy with
# four forwards in flight on the shared-chip table, and the same table with one lane as the control.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python scripts/r6_lanes_lab.py maskdiag > gpurun_out/r6_lab_maskdiag.txt 2> gpurun_out/r6_lab_maskdiag.err; echo "maskdiag rc $?"; cat gpurun_out/r6_lab_maskdiag.txt | cut -c1-200
timeout 900 python scripts/r6_lanes_lab.py controls > gpurun_out/r6_lab_controls.txt 2> gpurun_out/r6_lab_controls.err; echo "controls rc $?"; tail -12 gpurun_out/r6_lab_controls.txt; tail -3 gpurun_out/r6_lab_controls.err
# clocks and power while four lanes replay forwards for ~15 s
(for i in $(seq 1 14); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr '\n' ' '; echo; sleep 1; done) > gpurun_out/r6_power_under_4_lanes.txt &
SMI=$!
timeout 300 python scripts/lanes_replay.py 4 2500 > gpurun_out/r6_power_run.txt 2>&1
wait $SMI; cat gpurun_out/r6_power_under_4_lanes.txt | cut -c1-220
cd /tmp
for cfg in "4 6 4" "1 24 4" "1 24 1"; do
  set -- $cfg; tag=l$1_t$3
  rm -rf /tmp/pmc_$tag; mkdir -p /tmp/pmc_$tag
  i=0
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$tag/pass_$i -o p -- python $R/scripts/lanes_replay.py $1 $2 $3 > /tmp/pmc_$tag/pass_$i.log 2>&1 || tail -3 /tmp/pmc_$tag/pass_$i.log
    if [ "$tag" != "l4_t4" ] && [ $i -ge 2 ]; then break; fi   # (controls: fabric bytes only)
  done
  python $R/scripts/lanes_pmc_summary.py /tmp/pmc_$tag $1 $2 $R/gpurun_out/r6_pmc_$tag.json $R/gpurun_out/r6_pmc_$tag.txt > $R/gpurun_out/r6_pmc_$tag.log 2>&1; tail -40 $R/gpurun_out/r6_pmc_$tag.log | cut -c1-200
done
# kernel trace (no counters) of the same replay: do the lanes overlap as in the bench?
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python $R/scripts/lanes_replay.py 4 40 > /tmp/kt.log 2>&1
python - <<'PY' > $R/gpurun_out/r6_lanes_replay_trace_overlap.txt
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(t in r["Kernel_Name"] for t in ("igemm", "mlp_kernel", "hblock", "xblock", "attn_", "gn_", "ddim_step"))]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
n = len(iv); iv = iv[n // 4:]   # (skip the warm-up quarter)
ssum = sum(e - s for s, e in iv); union = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: union += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
union += ce - cs
nf = sum(1 for r in rows if "ddim_step" in r["Kernel_Name"]) * 3 // 4
print("kernels %d, sum of durations %.2f ms, union busy %.2f ms, avg kernels running %.2f, forwards (ddim_step kernels) ~%d, union per forward %.3f ms" % (len(iv), ssum / 1e6, union / 1e6, ssum / union, nf, union / 1e6 / max(1, nf)))
PY
cat $R/gpurun_out/r6_lanes_replay_trace_overlap.txt
