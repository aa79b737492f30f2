#!/bin/bash
# PMC passes on the patch kernel: pc_pmc.sh SHAPE CFG  -> gpurun_out/pc_pmc_SHAPE_CFG.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; S=$1; export UPK_PC_CFGS=$2
out=$R/gpurun_out/pc_pmc_${S}_$2.txt; : > $out
cd /tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
            "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc -o p -- python $R/scripts/pc_bench.py $S > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if "pconv_kernel" not in r["Kernel_Name"]: continue
    a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, v in agg.items():
    print("%-28s per-launch %14.1f  (n=%d)" % (k, v[1] / v[0], v[0]))
PY
done
cat $out
