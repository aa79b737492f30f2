#!/bin/bash
# PMC counter passes on single-shape micro benchmarks. Output: gpurun_out/pmc_*.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { # name args...
  name=$1; shift
  python $R/scripts/one_conv.py "$@" 2>&1 | grep shape
  for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
              "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
              "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES"; do
    rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc -o p -- python $R/scripts/one_conv.py "$@" > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
    python - "$f" "$name" >> $R/gpurun_out/pmc_$name.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if "igemm_kernel" not in r["Kernel_Name"]: continue
    a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, v in agg.items():
    print("%-28s per-launch %14.1f  (n=%d)" % (k, v[1] / v[0], v[0]))
PY
  done
  cat $R/gpurun_out/pmc_$name.txt
}
rm -f $R/gpurun_out/pmc_*.txt.tmp
: > $R/gpurun_out/pmc_convA.txt; : > $R/gpurun_out/pmc_convA_nosplit.txt; : > $R/gpurun_out/pmc_gemmE.txt; : > $R/gpurun_out/pmc_vae.txt
CFG_2722k2=15; CFG_1722k2=17; CFG_2222k4=7; CFG_4422k2=1
run convA 8 32 32 224 224 3 $CFG_2722k2 4 20
run convA_nosplit 8 32 32 224 224 3 $CFG_1722k2 1 20
run gemmE 1 8192 1 224 224 1 $CFG_2222k4 1 20
run vae 8 128 128 512 512 3 $CFG_4422k2 1 3
