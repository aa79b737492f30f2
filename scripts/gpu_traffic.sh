#!/bin/bash
# HBM-side traffic of the igemm kernels per launch: separate PMC passes (FETCH_SIZE, WRITE_SIZE) over N replays of
# the captured UNet forward -> gpurun_out/r04_igemm_traffic.json (copy to profiles/; bench.py reads it from there).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
N=4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/scripts/fwd_replay.py 32 32 $N > /tmp/pmc_$c.log 2>&1 || tail -3 /tmp/pmc_$c.log
done
python - $N "${GRAFT_COMMIT:-$(cat $R/.commit 2>/dev/null || echo unknown)}" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r04_igemm_traffic.json
import csv, glob, sys, json
N = int(sys.argv[1])
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    ig = [r for r in rows if "igemm" in r["Kernel_Name"] or "mlp_kernel" in r["Kernel_Name"] or "hblock_kernel" in r["Kernel_Name"] or "xblock_kernel" in r["Kernel_Name"]]
    main = [r for r in ig if "reduce" not in r["Kernel_Name"]]
    # the library's own per-forward kernels (the run also builds the model: weight packing, torch fills, rocBLAS)
    fwd = [r for r in rows if any(t in r["Kernel_Name"] for t in ("igemm", "mlp_kernel", "hblock_kernel", "xblock_kernel", "attn_", "gn_", "layernorm_kernel", "ddim_step"))]
    raw[c] = dict(kb_igemm=sum(float(r["Counter_Value"]) for r in ig), launches=len(main),
                  kb_all=sum(float(r["Counter_Value"]) for r in fwd))
L = raw["FETCH_SIZE"]["launches"]
fetch, write = raw["FETCH_SIZE"]["kb_igemm"] / L, raw["WRITE_SIZE"]["kb_igemm"] / L
print(json.dumps({
    "round": 4, "commit": sys.argv[2],
    "kernel_sources_sha256": open(__import__("os").environ["GRAFT_REPO_ROOT"] + "/upgpt_amd/libupk.so.sha256").read().strip(),
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python scripts/fwd_replay.py 32 32 %d (scripts/gpu_traffic.sh)" % N,
    "kernel_class": "igemm_ws_kernel<*> + igemm_kernel<*> + igemm_as_kernel<*> + mlp_kernel<*> + hblock_kernel<*> + xblock_kernel<*> + igemm_reduce[_gn|_gnapply]_kernel; per conv/GEMM launch incl. its reduce pass",
    "launches": L, "fetch_size_kb_per_launch_raw": fetch, "write_size_kb_per_launch_raw": write,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE as reported; KB -> x1024; Infinity-Cache hits are included (fabric-side counters)",
    "bytes_per_launch": (2 * fetch + write) * 1024.0,
    "all_kernels_bytes_per_forward": (2 * raw["FETCH_SIZE"]["kb_all"] + raw["WRITE_SIZE"]["kb_all"]) * 1024.0 / N,
    "all_kernels_note": "conv/GEMM + attention + GroupNorm + LayerNorm + sampler-step kernels of the %d replayed forwards (plus the ~40 step-invariant prep launches of the run); model-setup kernels (weight packing, fills) excluded" % N,
}, indent=1))
PY
