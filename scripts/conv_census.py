"""Per-shape census of the UNet step's conv/GEMM launches: count, tuned time, roofline time.

    python scripts/conv_census.py [H W]     (GPU; reads the committed tuning cache)
"""
import collections
import contextlib
import io
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd  # noqa: E402
from upgpt_amd import synth  # noqa: E402
from upgpt_amd.engine import TUNE_CACHE  # noqa: E402
from upgpt_amd import _lib  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model)
model = model.cuda()
pl = model.model.diffusion_model.plan(8, H, W, 87, 50, "sampler")
lib = _lib.load_library()
cnt = collections.Counter(k for _, k in pl.convs)
rows = []
for key, n in cnt.items():
    m = re.match(r"M(\d+)_N(\d+)_C(\d+)\+(\d+)_k(\d)s(\d)_f([0-9a-f]+)_r", key)
    M, N, C1, C2, ks = int(m[1]), int(m[2]), int(m[3]), int(m[4]), int(m[5])
    K = (C1 + C2) * ks * ks
    e = TUNE_CACHE.get(key)
    fl = 2.0 * M * N * K
    by = 2.0 * (M * (C1 + C2) + N * K + M * N)
    ideal = max(fl / 2.5e15, by / 6.0e12) * 1e6
    rows.append((n * e[2], n, key, e, fl, ideal))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("%d launches, %d shapes, tuned sum %.0f us" % (sum(cnt.values()), len(cnt), tot))
print("%7s %3s %-38s %-14s %2s %7s %7s %6s" % ("tot_us", "n", "shape", "cfg", "sk", "us", "ideal", "TF/s"))
for t, n, key, e, fl, ideal in rows:
    print("%7.1f %3d %-38s %-14s %2d %7.2f %7.2f %6.0f" % (
        t, n, key, lib.upk_conv_config_name(e[0]).decode(), e[1], e[2], ideal, fl / e[2] * 1e-6))
