#!/bin/bash
# per-op kernel timeline of the VAE decoder (scripts/op_trace.py, OT_VAE=1) -> gpurun_out/ot_vae_<H>x<W>.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
H=${1:-32}; W=${2:-32}
cd /tmp && rm -rf /tmp/otv && OT_VAE=1 OT_LABELS=$R/gpurun_out/otv_labels.json timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/otv -o ot -- python $R/scripts/op_trace.py run $H $W > $R/gpurun_out/otv_run.log 2>&1
cd $R
f=$(find /tmp/otv -name "*kernel_trace.csv" | head -1)
python scripts/op_trace.py parse "$f" gpurun_out/otv_labels.json > gpurun_out/ot_vae_${H}x${W}.txt 2> gpurun_out/otv_parse.err
tail -12 gpurun_out/ot_vae_${H}x${W}.txt
