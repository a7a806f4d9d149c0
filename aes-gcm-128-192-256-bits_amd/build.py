"""Build libaesgcm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# AESGCM_LIB selects an alternative build of the same sources (A/B experiments with other -D settings)
SO = os.environ.get("AESGCM_LIB") or os.path.join(HERE, "libaesgcm_hip.so")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    srcs = [os.path.join(CSRC, f) for f in ("aesgcm_kernels.hip", "aesgcm_comm.hip", "aesgcm_dev.h", "Makefile")]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "aesgcm.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, quiet=True):
    """Compile csrc/*.hip -> libaesgcm_hip.so.  Raises if hipcc is unavailable and no .so exists."""
    if os.environ.get("AESGCM_LIB"):
        return SO
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        if os.path.exists(SO):
            return SO        # prebuilt library travelling with the tree (GPU box without sources newer than it)
        raise RuntimeError("hipcc not found at %s and no prebuilt %s" % (hipcc, SO))
    cmd = ["make", "-C", CSRC, "HIPCC=" + hipcc] + (["-B"] if force else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        raise RuntimeError("building libaesgcm_hip.so failed:\n" + out.stdout[-4000:])
    if not quiet:
        print(out.stdout)
    return SO
