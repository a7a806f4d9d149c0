"""Where does the time of the first RCCL communicator go on a fresh box?  (round 2: the one-rank communicator test took
285 s.)  Prints a timestamp after each step; run with NCCL_DEBUG=INFO to see RCCL's own log beside it."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T0 = time.time()


def stamp(what):
    print("[%7.2f s] %s" % (time.time() - T0, what), flush=True)


if "--cat" in sys.argv:
    n = 0
    with open("/opt/rocm/lib/librccl.so.1", "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            n += len(b)
    stamp("read librccl.so.1 sequentially: %d MiB" % (n >> 20))
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402
lib.load()
stamp("libaesgcm_hip.so loaded")
lib.device_count()
stamp("hip device count")
ctypes.CDLL("librccl.so.1", mode=ctypes.RTLD_GLOBAL)
stamp("dlopen librccl.so.1")
uid = lib.comm_unique_id()
stamp("ncclGetUniqueId")
c = lib.Comm(uid, 1, 0, device=0)
stamp("ncclCommInitRank (1 rank)")
a, b = lib.DeviceBuffer(32), lib.DeviceBuffer(32)
a.upload(bytes(range(32)))
c.allgather_dev(a.ptr, b.ptr, 32)
c.barrier()
stamp("first all-gather + barrier")
c.allgather_dev(a.ptr, b.ptr, 32)
c.barrier()
stamp("second all-gather + barrier")
c.close()
stamp("comm destroyed")
uid = lib.comm_unique_id()
c = lib.Comm(uid, 1, 0, device=0)
stamp("second communicator in the same process")
c.close()
