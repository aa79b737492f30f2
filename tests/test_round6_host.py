"""Host logic added in round 6 (no GPU): CU-slot masks of the lane streams, the shared weight packs of the CLIP towers, the
next-weight prefetch links of a step program, and the two reducers the bench's roofline block reads its committed profiler
summaries from (scripts/trace_frac.py, scripts/lanes_pmc_summary.py)."""
import csv
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cu_slot_masks_partition_the_chip_and_touch_every_xcd():
    """lanes.cu_slot_mask_words: mask bit b = CU slot b // 8 of XCD b % 8 (measured, profiles/r06_cu_mask_semantics.txt).
    n partitions of 32 / n slots are disjoint, cover all 256 bits, and every partition has bits in EVERY XCD — an XCD whose
    bits are all zero would be unrestricted on this driver."""
    from upgpt_amd.lanes import N_CU, N_XCD, cu_slot_mask_words
    for n in (1, 2, 4, 8, 16, 32):
        masks = []
        for l in range(n):
            w = cu_slot_mask_words(set(range(l * 32 // n, (l + 1) * 32 // n)))
            assert len(w) == N_CU // 32 and all(0 <= x < (1 << 32) for x in w)
            bits = {32 * i + b for i, x in enumerate(w) for b in range(32) if x >> b & 1}
            assert len(bits) == N_CU // n
            assert {b % N_XCD for b in bits} == set(range(N_XCD))
            masks.append(bits)
        assert set().union(*masks) == set(range(N_CU)) and sum(len(m) for m in masks) == N_CU


def test_shared_packs_build_each_operand_once_and_count_fresh_ones():
    from upgpt_amd.packing import SharedPacks

    class FakePacker:
        def __init__(self):
            self.calls = []

        def pack(self, names, **kw):
            self.calls.append(("pack", names if isinstance(names, str) else tuple(names), tuple(sorted(kw.items()))))
            return object()

        def vec(self, name):
            self.calls.append(("vec", name))
            return object()

        other = "plain"

    sh = SharedPacks()
    a, b = FakePacker(), FakePacker()  # two lanes' packers over the same weights
    pa, pb = sh.wrap(a), sh.wrap(b)
    w1 = pa.pack("l0.fc1", ln="l0.ln2")
    assert pb.pack("l0.fc1", ln="l0.ln2") is w1 and pa.pack(["q", "k", "v"], n_out=8) is pb.pack(["q", "k", "v"], n_out=8)
    assert pa.pack("l0.fc1") is not w1  # (other keyword arguments = another operand)
    assert pa.vec("g") is pb.vec("g") and pa.other == "plain"
    assert len(a.calls) == 4 and b.calls == []  # the second lane built nothing
    assert sh.take_fresh() == 4 and sh.take_fresh() == 0


def test_prefetch_links_follow_launch_order_wrap_and_respect_the_scope(monkeypatch):
    """Emitter.link_weight_prefetch on a fake step program: conv k points at conv k + 1's packed weight, the last at the
    first, ops without a weight are skipped, a conv never prefetches its own weight, the byte cap holds; "auto" arms the
    links only while the calling thread builds for one batch in flight (_lib.concurrency)."""
    from upgpt_amd import _lib as L
    from upgpt_amd import knobs
    from upgpt_amd.emitter import Emitter

    def desc(ptr, nbytes):
        d = types.SimpleNamespace(w_packed=ptr, pf_next=None, pf_bytes=0)
        d._w_bytes = nbytes
        return d

    ds = [desc(0x1000, 4096), desc(0x9000, 1 << 20), desc(0x9000, 1 << 20), desc(0x5000, 64 << 20)]
    prog = types.SimpleNamespace(meta=[ds[0], None, ds[1], None, None, ds[2], ds[3]])
    em = Emitter.__new__(Emitter)
    monkeypatch.setattr(knobs, "WEIGHT_PREFETCH", "auto")
    monkeypatch.setattr(knobs, "WEIGHT_PREFETCH_AHEAD", 1)
    monkeypatch.setattr(knobs, "WEIGHT_PREFETCH_MAX", 32 << 20)
    assert em.link_weight_prefetch(prog) == 3
    assert (ds[0].pf_next, ds[0].pf_bytes) == (0x9000, 1 << 20)
    assert ds[1].pf_next is None  # (the next launch reads the SAME weight: nothing to warm)
    assert (ds[2].pf_next, ds[2].pf_bytes) == (0x5000, 32 << 20)  # capped
    assert (ds[3].pf_next, ds[3].pf_bytes) == (0x1000, 4096)  # wraps to the first launch of the next step
    for d in ds:
        d.pf_next, d.pf_bytes = None, 0
    with L.shared_chip(4):
        assert em.link_weight_prefetch(prog) == 0 and all(d.pf_next is None for d in ds)
    monkeypatch.setattr(knobs, "WEIGHT_PREFETCH", "0")
    assert em.link_weight_prefetch(prog) == 0
    monkeypatch.setattr(knobs, "WEIGHT_PREFETCH", "1")
    with L.shared_chip(4):
        assert em.link_weight_prefetch(prog, wrap=False) == 2 and ds[3].pf_next is None


def _write_csv(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)


def test_trace_frac_tells_the_serial_phase_from_the_lanes_phase(tmp_path):
    """scripts/trace_frac.py on a synthetic kernel trace: 200 ms of one kernel at a time (one forward per 10 kernels of 50 us)
    followed by 600 ms with three kernels running at any time; frac = forwards x GF / union / 2.5 PF per phase."""
    rows, t = [], 0
    for i in range(4000):
        rows.append(dict(Kernel_Name="ddim_step_kernel" if i % 10 == 0 else "igemm_ws_kernel<1>", Start_Timestamp=t, End_Timestamp=t + 50000))
        t += 50000
    for i in range(12000):
        rows.append(dict(Kernel_Name="ddim_step_kernel" if i % 10 == 0 else "igemm_ws_kernel<4>", Start_Timestamp=t, End_Timestamp=t + 150000))
        t += 50000
    p = tmp_path / "trace.csv"
    _write_csv(p, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "trace_frac.py"), str(p), str(tmp_path / "o.json"),
                          str(tmp_path / "o.txt"), "1000.0"], capture_output=True, text=True, check=True)
    r = json.loads(out.stdout)
    assert r["serial_windows"] >= 1 and r["steady_windows"] >= 4
    assert abs(r["kernels_running_while_busy"] - 3.0) < 0.05 and abs(r["serial_phase"]["kernels_running_while_busy"] - 1.0) < 0.01
    # one forward (1000 GF) per 0.5 ms in both phases -> 2 PFLOP/s -> 0.8 of the 2.5 PFLOP/s peak
    assert abs(r["frac_from_trace"] - 0.8) < 0.02 and abs(r["serial_phase"]["frac_from_trace"] - 0.8) < 0.02
    table = (tmp_path / "o.txt").read_text().splitlines()
    assert table[0].startswith("# {") and len(table) > 5


def test_lanes_pmc_summary_applies_the_fetch_correction_and_finds_the_dominant_kernel(tmp_path):
    """scripts/lanes_pmc_summary.py: FETCH_SIZE (KB) x 2 x 1024 + WRITE_SIZE (KB) x 1024 per lane-forward, the conv/GEMM class
    apart from the other forward kernels, non-forward kernels ignored, the dominant (kernel, grid) per launch."""
    dom = "void (anonymous namespace)::igemm_ws_kernel<4, 7, 2, 2, 2, 3, false, false>(upkd::IgemmArgs)"

    def rows(counter, per_kernel):
        out = []
        for name, grid, val, n in per_kernel:
            for _ in range(n):
                out.append(dict(Kernel_Name=name, Grid_Size=grid, Counter_Name=counter, Counter_Value=val))
        return out
    kernels = [(dom, 32768, 100.0, 8), ("void (anonymous namespace)::gn_apply_kernel(x)", 8192, 10.0, 8), ("at::native::copy_kernel", 64, 999.0, 8)]
    for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE"), 1):
        d = tmp_path / ("pass_%d" % i) / "x"
        d.mkdir(parents=True)
        _write_csv(d / "p_counter_collection.csv", rows(c, kernels))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lanes_pmc_summary.py"), str(tmp_path), "4", "2",
                          str(tmp_path / "o.json"), str(tmp_path / "o.txt")], capture_output=True, text=True, check=True)
    r = json.load(open(tmp_path / "o.json"))
    assert r["per_lane_forward"]["FETCH_SIZE"] == pytest.approx((800 + 80) * 2 * 1024 / 8)
    assert r["per_lane_forward_conv_gemm_class"]["WRITE_SIZE"] == pytest.approx(800 * 1024 / 8)
    assert r["fabric_bytes_per_lane_forward"] == pytest.approx((880 * 2 + 880) * 1024 / 8)
    assert r["dominant_kernel"]["kernel"].startswith("igemm_ws_kernel<4, 7, 2, 2, 2, 3") and r["dominant_kernel"]["launches"] == 8
    assert r["dominant_kernel"]["per_launch"]["FETCH_SIZE"] == pytest.approx(100 * 2 * 1024)
    assert out.stdout.strip().startswith("{")
