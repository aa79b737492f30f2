#!/bin/bash
# per-op kernel timeline of the UNet forward (scripts/op_trace.py) -> gpurun_out/ot_table_<H>x<W>.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
H=${1:-32}; W=${2:-32}
cd /tmp && rm -rf /tmp/ot && OT_LABELS=$R/gpurun_out/ot_labels.json timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ot -o ot -- python $R/scripts/op_trace.py run $H $W > $R/gpurun_out/ot_run.log 2>&1
cd $R
f=$(find /tmp/ot -name "*kernel_trace.csv" | head -1)
python scripts/op_trace.py parse "$f" gpurun_out/ot_labels.json > gpurun_out/ot_table_${H}x${W}.txt 2> gpurun_out/ot_parse.err
tail -12 gpurun_out/ot_table_${H}x${W}.txt
