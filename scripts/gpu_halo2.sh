#!/bin/bash
# halo-patch kernel: phase ablation (dev build) and PMC passes on single shapes
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
{
for abl in 0 0x20000 0x40000 0x80000 0x60000 0xe0000; do
  echo "UPK_ABLATE=$abl"
  UPK_LIB=$R/upgpt_amd/libupk_dev.so UPK_ABLATE=$abl EPI=bias,res python scripts/one_conv.py 8 32 32 224 224 3 81 1 20 2>&1 | grep -E "shape|rror"
  UPK_LIB=$R/upgpt_amd/libupk_dev.so UPK_ABLATE=$abl EPI=bias,res python scripts/one_conv.py 8 16 16 448 448 3 83 1 20 2>&1 | grep -E "shape|rror"
done
} | tee gpurun_out/halo_ablate.log
cd /tmp
pmc() { # name args...
  name=$1; shift
  : > $R/gpurun_out/pmc_$name.txt
  EPI=bias,res python $R/scripts/one_conv.py "$@" 2>&1 | grep shape >> $R/gpurun_out/pmc_$name.txt
  for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
              "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
              "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES"; do
    rm -rf /tmp/pmc; EPI=bias,res timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc -o p -- python $R/scripts/one_conv.py "$@" > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
    python - "$f" >> $R/gpurun_out/pmc_$name.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if "halo_conv" not in r["Kernel_Name"] and "igemm" not in r["Kernel_Name"]: continue
    a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, v in agg.items():
    print("%-28s per-launch %14.1f  (n=%d)" % (k, v[1] / v[0], v[0]))
PY
  done
  cat $R/gpurun_out/pmc_$name.txt
}
pmc halo_hc7p4_conv224 8 32 32 224 224 3 81 1 20
