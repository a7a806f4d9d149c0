"""Build libaesgcm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# AESGCM_LIB selects an alternative build of the same sources (A/B experiments with other -D settings)
SO = os.environ.get("AESGCM_LIB") or os.path.join(HERE, "libaesgcm_hip.so")
# the same sources with -DAESGCM_DEBUG_KNOBS: exports aesgcm_debug_force_shape as well (include/aesgcm_debug.h; tests and profiling scripts)
SO_DEBUG = os.environ.get("AESGCM_LIB_DEBUG") or os.path.join(HERE, "libaesgcm_hip_dbg.so")


def needs_build():
    if not os.path.exists(SO) or not os.path.exists(SO_DEBUG):
        return True
    t = min(os.path.getmtime(SO), os.path.getmtime(SO_DEBUG))
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc")) or f == "Makefile"]
    srcs += [os.path.join(os.path.dirname(HERE), "include", f) for f in ("aesgcm.h", "aesgcm_debug.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, quiet=True):
    """Compile csrc/*.hip -> libaesgcm_hip.so.  Raises if hipcc is unavailable and no .so exists."""
    if os.environ.get("AESGCM_LIB"):
        return SO
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        if os.path.exists(SO) and os.path.exists(SO_DEBUG):
            return SO        # prebuilt library travelling with the tree (GPU box without sources newer than it)
        raise RuntimeError("hipcc not found at %s and no prebuilt %s" % (hipcc, SO))
    cmd = ["make", "-C", CSRC, "-j2", "HIPCC=" + hipcc] + (["-B"] if force else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        raise RuntimeError("building libaesgcm_hip.so failed:\n" + out.stdout[-4000:])
    if not quiet:
        print(out.stdout)
    return SO
