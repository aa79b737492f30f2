"""The tuning table: shape signature of a conv / GEMM launch -> measured (tile configuration, split-K)."""
import json
import os

from ._check import require


class TuneCache:
    """shape signature -> [cfg, splitk, best_us, default_us], measured on an MI355X by
    upk_conv_autotune and kept in-tree (upgpt_amd/tuned_gfx950.json) so that fresh processes
    start with tuned launches.  Unknown shapes fall back to the library's cost model."""

    def __init__(self, path=None):
        self.path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json")
        self.d = {}
        self.dirty = False
        self.names = None  # configuration names the indices in the file refer to ("__configs__"), resolved by bind()
        if os.path.exists(self.path):
            try:
                with open(self.path) as f:
                    self.d = json.load(f)
            except Exception:
                self.d = {}
        self.names = self.d.pop("__configs__", None)
        # other "__...__" keys are decisions that go with the table (e.g. "__unfuse_mlp_M__": row counts for which the fused
        # feed-forward tail loses to two launches tuned for a shared chip); kept apart from the shape entries
        self.meta = {k: self.d.pop(k) for k in list(self.d) if k.startswith("__")}
        self._bound = False

    def bind(self, lib):
        """Ties the stored configuration indices to THIS library's configuration list: entries are re-indexed by
        configuration name when the file records the names it was written with ("__configs__"), and entries whose
        configuration does not exist (any more) are dropped — they fall back to the cost model instead of pinning a
        different kernel or an index past the table."""
        if self._bound:
            return
        self._bound = True
        n = lib.upk_conv_num_configs()
        cur = [lib.upk_conv_config_name(i).decode() for i in range(n)]
        if self.names is not None and self.names != cur:
            idx = {nm: i for i, nm in enumerate(cur)}
            remap = {i: idx.get(nm, -1) for i, nm in enumerate(self.names)}
            for k in list(self.d):
                try:
                    c = remap.get(int(self.d[k][0]), -1)
                except (TypeError, ValueError, IndexError, KeyError):
                    c = -1  # (a malformed entry is dropped, never a reason to fail)
                if c < 0:
                    del self.d[k]
                else:
                    self.d[k][0] = c
        else:
            for k in list(self.d):
                try:
                    ok = 0 <= int(self.d[k][0]) < n
                except (TypeError, ValueError, IndexError, KeyError):
                    ok = False
                if not ok:
                    del self.d[k]
        self.names = cur

    def get(self, key):
        return self.d.get(key)

    def put(self, key, cfg, sk, best_us, dflt_us):
        # indices written now refer to THIS library's configuration list: the stored ones must have been re-indexed first
        require(self._bound, "TuneCache.put before bind(lib): stored and new configuration indices would mix", RuntimeError)
        self.d[key] = [int(cfg), int(sk), round(float(best_us), 2), round(float(dflt_us), 2)]
        self.dirty = True
        return self.d[key]

    def save(self, path=None):
        out = dict(self.d)
        out.update(self.meta)
        if self.names is not None:
            out["__configs__"] = self.names
        with open(path or self.path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
        self.dirty = False


TUNE_CACHE = TuneCache(os.environ.get("UPGPT_TUNE_FILE") or None)
# Overlay for plans built while several batches are in flight on the device (execution lanes, lanes.py): the choices of
# an in-situ pass whose objective is the time per forward of THREE concurrent forwards (scripts/tune_insitu.py with
# INSITU_LANES=3) — with the chip shared, a launch is priced by the CU time it takes, not by its latency alone.  Only
# the shapes that pass changed; everything else falls through to TUNE_CACHE.
TUNE_CACHE_LANES = TuneCache(os.environ.get("UPGPT_TUNE_FILE_LANES") or
                             os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950_lanes.json"))
