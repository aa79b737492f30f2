"""Builds libupk.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

No torch / pybind dependency: the library is a plain C-ABI shared object
(include/upk.h) loaded through ctypes by upgpt_amd/_lib.py.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("UPK_LIB") or os.path.join(HERE, "libupk.so")  # (UPK_LIB + UPK_CXXFLAGS: dev builds)
SOURCES = ["igemm.hip", "bigtile.hip", "astat.hip", "mlp.hip", "xblock.hip", "attention.hip", "norm.hip", "misc.hip"]
# per-file flags.  attention.hip: MFMA results straight into arch VGPRs — the softmax between the two matmuls reads
# every score with VALU instructions, and with the accumulators in AGPRs 112 of ~600 issue slots per 64-key tile were
# v_accvgpr moves (the kernels use < 128 registers, there is nothing to gain from the AGPR file)
# -fno-honor-nans -mno-amdgpu-ieee: fmaxf on MFMA results otherwise quiets every operand first (v_max x, x) — two of
# three instructions of the running-maximum tree; the kernels never produce a NaN (masked scores are -inf, every tile
# has a visible key per row, so no inf - inf) and infinities keep their meaning
FILE_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-mno-amdgpu-ieee"]}
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libupk.so cannot be built")


def _source_hash():
    """SHA-256 over every file of csrc/ and include/upk.h (names + contents) and the compile flags."""
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "upk.h")]
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    h.update((ARCH + " " + os.environ.get("UPK_CXXFLAGS", "") + repr(sorted(FILE_FLAGS.items()))).encode())
    return h.hexdigest()


def _stale():
    """The library is rebuilt when it is missing, when UPK_FORCE_BUILD=1, or when the hash recorded next to it
    (libupk.so.sha256, written by the build that produced it) differs from the hash of the sources — a shipped binary
    that does not come from THESE sources is never reused (modification times say nothing on a fresh checkout)."""
    if os.environ.get("UPK_FORCE_BUILD", "0") == "1" or not os.path.exists(OUT):
        return True
    try:
        with open(OUT + ".sha256") as f:
            return f.read().strip() != _source_hash()
    except OSError:
        return True


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link libupk.so. Returns its path."""
    if not force and not _stale():
        return OUT
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build" + ("_dev" if os.environ.get("UPK_LIB") else ""))
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
             "-Wno-unused-result", "-I", os.path.join(HERE, "..", "include")]
    flags += os.environ.get("UPK_CXXFLAGS", "").split()  # dev builds, e.g. -DUPK_TIMELINE (scripts/timeline.py)

    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(HERE, "..", "include", "upk.h")]

    def obj_hash(src):  # an object is reused only when its source, every header and its flags are unchanged
        h = hashlib.sha256()
        for path in [os.path.join(CSRC, src)] + headers:
            with open(path, "rb") as f:
                h.update(f.read())
        h.update(repr(flags + FILE_FLAGS.get(src, [])).encode())
        return h.hexdigest()

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        want = obj_hash(src)
        try:
            with open(obj + ".sha256") as f:
                if not force and os.path.exists(obj) and f.read().strip() == want and os.environ.get("UPK_FORCE_BUILD", "0") != "1":
                    return obj
        except OSError:
            pass
        cmd = [hipcc] + flags + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[upgpt_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(obj + ".sha256", "w") as f:
            f.write(want + "\n")
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT + ".tmp"] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(OUT + ".tmp", OUT)
    with open(OUT + ".sha256", "w") as f:
        f.write(_source_hash() + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
