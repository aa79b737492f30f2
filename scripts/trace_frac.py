"""MFMA-roofline fraction of a run recomputed from its rocprofv3 kernel trace alone (`frac_from_trace`):
   Sum of algorithmic FLOPs of the UNet forwards executed  /  union of the kernel intervals  /  2.5 PFLOP/s.
   python scripts/trace_frac.py <kernel_trace.csv> <out.json> <out.txt> [gflop per forward = 728.23898112] [window ms = 100]
A forward is counted by its ddim_step kernel (one per UNet evaluation of the sampler).  The run is cut into windows; the
steady-state windows (device busy >= 90 % of the window and at least one forward) are the timed configuration; everything
is written per window so that the figure can be recomputed by hand from the committed table."""
import csv, json, re, sys
path, out_json, out_txt = sys.argv[1:4]
GF = float(sys.argv[4]) if len(sys.argv) > 4 else 728.23898112
W = int(float(sys.argv[5]) * 1e6) if len(sys.argv) > 5 else 100_000_000
PEAK = 2.5e15
rows = list(csv.DictReader(open(path)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "ddim_step" in r["Kernel_Name"],
             any(t in r["Kernel_Name"] for t in ("igemm", "mlp_kernel", "hblock_kernel", "xblock_kernel"))) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
wins = []
w = t0
while w < t1:
    sel = [(max(s, w), min(e, w + W), d, g, s) for s, e, d, g in iv if e > w and s < w + W]
    if sel:
        ssum = sum(e - s for s, e, _, _, _ in sel)
        gsum = sum(e - s for s, e, _, g, _ in sel if g)
        union, cs, ce = 0, None, None
        for s, e, _, _, _ in sel:
            if ce is None or s > ce:
                if ce is not None:
                    union += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        union += ce - cs
        fw = sum(1 for _, _, d, _, s0 in sel if d and s0 >= w)  # (a forward belongs to the window its step kernel starts in)
        wins.append(dict(start_ms=(w - t0) / 1e6, kernels=len(sel), sum_ms=ssum / 1e6, gemm_sum_ms=gsum / 1e6, union_ms=union / 1e6, forwards=fw, busy=union / W))
    w += W
# the bench command runs the steps one batch at a time first (`serial`: one kernel at a time, sum of durations ~ union), then with
# the lanes in flight (sum / union ~ 3): the two phases are told apart by that ratio, window by window
busy = [x for x in wins if x["busy"] >= 0.9 and x["forwards"] > 0]
steady = [x for x in busy if x["sum_ms"] / x["union_ms"] >= 1.5]
serial = [x for x in busy if x["sum_ms"] / x["union_ms"] < 1.2]
res = {"gflop_per_forward": GF, "peak_tflops": PEAK / 1e12, "window_ms": W / 1e6, "windows": len(wins), "steady_windows": len(steady),
       "serial_windows": len(serial)}


def phase(sel):
    fw = sum(x["forwards"] for x in sel)
    un = sum(x["union_ms"] for x in sel)
    ss = sum(x["sum_ms"] for x in sel)
    return {"forwards": fw, "union_busy_ms": un, "sum_of_kernel_durations_ms": ss, "kernels_running_while_busy": ss / un,
            "union_ms_per_forward": un / fw, "frac_from_trace": fw * GF * 1e9 / (un * 1e-3) / PEAK}


if steady:
    res.update(phase(steady))
    res["note"] = ("windows with the lanes in flight (device busy >= 90 %, sum of kernel durations / union >= 1.5); all their kernels "
                   "count as busy time (VAE decode, layout converts and the gather included), only UNet FLOPs count as work; the "
                   "run is slower under the tracer than the untraced bench line (per-dispatch profiler overhead)")
if serial:
    res["serial_phase"] = phase(serial)
json.dump(res, open(out_json, "w"), indent=1)
with open(out_txt, "w") as f:
    f.write("# %s\n# window_start_ms kernels sum_of_durations_ms conv_gemm_sum_ms union_busy_ms forwards busy_share\n" % json.dumps({k: v for k, v in res.items() if k != "note"}))
    for x in wins:
        f.write("%10.1f %7d %10.2f %10.2f %10.2f %5d %7.3f\n" % (x["start_ms"], x["kernels"], x["sum_ms"], x["gemm_sum_ms"], x["union_ms"], x["forwards"], x["busy"]))
print(json.dumps(res))
