#!/bin/bash
# Round-5 evidence session: PMC passes over the replayed forward for the dominant conv and for the
# kernels that had no counters yet (mlp / hblock / xblock / attn_lds), the L2 request bytes of the dominant kernel, the
# fabric traffic summary bench.py reads.  Output: gpurun_out/r05_*.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd /tmp
KERNELS="igemm_ws_kernel<1, 7, 4, 1, 4, 3 mlp_kernel hblock_kernel xblock_kernel attn_lds_kernel"
: > $R/gpurun_out/r05_pmc_forward_kernels.txt
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
            "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf /tmp/p5; timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/p5 -o p -- python $R/scripts/fwd_replay.py 32 32 4 > /tmp/p5.log 2>&1 || tail -3 /tmp/p5.log
  f=$(find /tmp/p5 -name "*counter_collection.csv" | head -1)
  echo "## pass: $pass" >> $R/gpurun_out/r05_pmc_forward_kernels.txt
  python $R/scripts/pmc_fwd_by_kernel.py "$f" "igemm_ws_kernel<1, 7, 4, 1, 4, 3" mlp_kernel hblock_kernel xblock_kernel attn_ >> $R/gpurun_out/r05_pmc_forward_kernels.txt
  if echo "$pass" | grep -q TCC_HIT; then cp "$f" /tmp/p5_tcc.csv; fi
done
# fabric traffic (FETCH / WRITE) + the dominant kernel's L2 request bytes -> the summary bench.py reads
N=4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/scripts/fwd_replay.py 32 32 $N > /tmp/pmc_$c.log 2>&1 || tail -3 /tmp/pmc_$c.log
done
python - $N "${GRAFT_COMMIT:-$(cat $R/.commit 2>/dev/null || echo unknown)}" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r05_igemm_traffic.json
import csv, glob, sys, json, os, collections, re
N = int(sys.argv[1])
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    ig = [r for r in rows if "igemm" in r["Kernel_Name"] or "mlp_kernel" in r["Kernel_Name"] or "hblock_kernel" in r["Kernel_Name"] or "xblock_kernel" in r["Kernel_Name"]]
    main = [r for r in ig if "reduce" not in r["Kernel_Name"]]
    fwd = [r for r in rows if any(t in r["Kernel_Name"] for t in ("igemm", "mlp_kernel", "hblock_kernel", "xblock_kernel", "attn_", "gn_", "layernorm_kernel", "ddim_step"))]
    raw[c] = dict(kb_igemm=sum(float(r["Counter_Value"]) for r in ig), launches=len(main), kb_all=sum(float(r["Counter_Value"]) for r in fwd))
L = raw["FETCH_SIZE"]["launches"]
fetch, write = raw["FETCH_SIZE"]["kb_igemm"] / L, raw["WRITE_SIZE"]["kb_igemm"] / L
# L2 requests of the dominant (kernel, grid): TCC_HIT_sum + TCC_MISS_sum, 128-byte requests
tcc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open("/tmp/p5_tcc.csv")):
    k = re.sub(r"\(anonymous namespace\)::|void |upkd::", "", r["Kernel_Name"]).split("(")[0][:52] + " g" + r["Grid_Size"]
    a = tcc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
dom = [k for k in tcc if k.startswith("igemm_ws_kernel<1, 7, 4, 1, 4, 3") and k.endswith("g131072")]
l2 = None
if dom:
    d = tcc[dom[0]]
    l2 = dict(kernel=dom[0], launches=int(d["TCC_HIT_sum"][0]),
              requests_per_launch=(d["TCC_HIT_sum"][1] + d["TCC_MISS_sum"][1]) / d["TCC_HIT_sum"][0],
              hit_rate=d["TCC_HIT_sum"][1] / (d["TCC_HIT_sum"][1] + d["TCC_MISS_sum"][1]))
    l2["bytes_per_launch"] = l2["requests_per_launch"] * 128.0
print(json.dumps({
    "round": 5, "commit": sys.argv[2],
    "kernel_sources_sha256": open(os.environ["GRAFT_REPO_ROOT"] + "/upgpt_amd/libupk.so.sha256").read().strip(),
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE (separate passes) -- python scripts/fwd_replay.py 32 32 %d (scripts/gpu_r5_evidence.sh)" % N,
    "kernel_class": "igemm_ws_kernel<*> + igemm_kernel<*> + igemm_as_kernel<*> + mlp_kernel<*> + hblock_kernel<*> + xblock_kernel<*> + igemm_reduce[_gn|_gnapply]_kernel; per conv/GEMM launch incl. its reduce pass",
    "launches": L, "fetch_size_kb_per_launch_raw": fetch, "write_size_kb_per_launch_raw": write,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE as reported; KB -> x1024; Infinity-Cache hits are included (fabric-side counters)",
    "bytes_per_launch": (2 * fetch + write) * 1024.0,
    "all_kernels_bytes_per_forward": (2 * raw["FETCH_SIZE"]["kb_all"] + raw["WRITE_SIZE"]["kb_all"]) * 1024.0 / N,
    "dominant_kernel_l2": l2,
    "dominant_kernel_l2_note": "TCC_HIT_sum + TCC_MISS_sum = 128-byte requests served by the XCDs' L2s for the (kernel, grid) rocprofv3 ranks first in the bench (the 3x3 224 -> 224 conv of the 32x32 level); per launch inside the replayed forward",
}, indent=1))
PY
cp $R/gpurun_out/r05_igemm_traffic.json $R/profiles/r05_igemm_traffic.json 2>/dev/null
tail -60 $R/gpurun_out/r05_pmc_forward_kernels.txt
