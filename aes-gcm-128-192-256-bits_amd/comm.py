"""Rank plumbing for the one-process-per-GPU launch of the sharded path -- no PyTorch.

`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py` only LAUNCHES the ranks (it sets RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT); nothing here imports torch.  The ranks of one node need two things:

* a way to hand rank 0's 128-byte RCCL unique id to the others: a file under /tmp whose name carries the launcher's
  pid (every rank has the same parent) and MASTER_PORT, written atomically, removed by rank 0 at exit;
* the exchange itself: `RcclExchange` = lib.Comm (ncclCommInitRank / ncclAllGather inside libaesgcm_hip.so).

`FileExchange` is the debug stand-in used when several ranks share ONE GPU (RCCL refuses duplicate devices): the
16-byte partials go through the host and a directory of small files.  It exists so that the multi-process
orchestration of the product path can be tested on a one-GPU box (tests/test_gpu_multiproc.py); it is never the
benchmarked path.
"""
import os
import struct
import sys
import tempfile
import time

from . import lib


def env_rank():
    """(rank, world, local_rank) from the launcher's environment; (0, 1, 0) when not launched as ranks."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def rendezvous_dir():
    """A directory every rank of this launch (and only of this launch) agrees on."""
    d = os.environ.get("AESGCM_RDZV_DIR")
    if d:
        return d
    key = "%s_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()   # memory-backed when possible
    return os.path.join(base, "aesgcm_rdzv_%d_%s" % (os.getuid(), key))


def _write_atomic(path, data):
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)


def _wait_read(path, nbytes, timeout):
    t0 = time.monotonic()
    while True:
        try:
            with open(path, "rb") as f:
                b = f.read()
            if len(b) == nbytes:
                return b
        except OSError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError("rendezvous: %s did not appear within %.0f s" % (path, timeout))
        time.sleep(0.005)


# How long a rank waits for the others.  The first RCCL call of a process loads a 573 MB library: 5 s with the file cached, minutes on a
# cold box (seen in round 3), and every rank of a node does it at the same time.  The waits below are upper bounds for that, sized to sit
# together (pre-flight + unique id + communicator) inside the driver's 1800 s per command.
WAIT_PREFLIGHT_S = 700.0      # every rank has loaded librccl and says so
WAIT_UNIQUE_ID_S = 300.0      # rank 0 (its library already loaded) publishes the unique id
WAIT_COMM_S = 600.0           # every rank is back from ncclCommInitRank


def share_bytes(name, data, rank, nbytes, timeout=WAIT_UNIQUE_ID_S):
    """rank 0 publishes `data` (nbytes) under `name`; every rank returns it."""
    d = rendezvous_dir()
    path = os.path.join(d, name)
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        _write_atomic(path, data)
        return data
    return _wait_read(path, nbytes, timeout)


def finish(rank, world, timeout=60.0):
    """Last call of a rank: says goodbye; rank 0 waits for every goodbye, then removes the rendezvous directory (nobody
    reads a file after its own goodbye, so nothing is pulled from under a reader)."""
    d = rendezvous_dir()
    try:
        os.makedirs(d, exist_ok=True)
        _write_atomic(os.path.join(d, "bye_%d" % rank), b"\x01")
        if rank != 0:
            return
        for r in range(world):
            try:
                _wait_read(os.path.join(d, "bye_%d" % r), 1, timeout)
            except TimeoutError:
                pass
        for f in os.listdir(d):
            try:
                os.unlink(os.path.join(d, f))
            except OSError:
                pass
        os.rmdir(d)
    except OSError:
        pass


def make_exchange(rank, world, device, prefer="rccl"):
    """The exchange for this launch.  `prefer="rccl"`: every rank tries the RCCL communicator and reports whether it came up;
    only if ALL ranks have one is it used -- otherwise every rank falls back to the file exchange (the payload is 16 bytes
    per message, so even that costs well under 1 % of a 16 GiB step) and the bench line says so.  This exists because the
    builder had no multi-GPU box: the first multi-rank RCCL run of this code is the driver's scaling run."""
    if prefer != "rccl":
        return FileExchange(rank, world, device)
    d = rendezvous_dir()
    os.makedirs(d, exist_ok=True)

    def agree(name, ok, timeout):
        """every rank publishes ok/not ok under `name`; -> the list of all ranks' answers (a rank that never answers counts as not ok)"""
        _write_atomic(os.path.join(d, "%s_%d" % (name, rank)), b"\x01" if ok else b"\x00")
        out = []
        for r in range(world):
            try:
                out.append(_wait_read(os.path.join(d, "%s_%d" % (name, r)), 1, timeout) == b"\x01")
            except TimeoutError:
                out.append(False)
        return out

    def complain(stage, err):
        # the library's own words (aesgcm_comm_last_error: RCCL's error string) on EVERY rank's stderr; with NCCL_DEBUG=WARN in the
        # environment (bench.py sets it for the ranks it starts) RCCL's warning lines are already there
        try:
            detail = lib.load().aesgcm_comm_last_error().decode()
        except Exception:                       # noqa: BLE001
            detail = ""
        sys.stderr.write("aesgcm comm rank %d/%d device %d: %s failed: %s | %s\n" % (rank, world, device, stage, err, detail))
        sys.stderr.flush()

    # pre-flight: can every rank load RCCL at all?  (a rank that cannot must not leave the others blocked inside ncclCommInitRank)
    ex, err = None, ""
    try:
        lib.comm_unique_id()
        pre = True
    except Exception as e:                      # noqa: BLE001
        pre, err = False, repr(e)
        complain("loading RCCL (ncclGetUniqueId)", err)
    oks = agree("rccl_pre", pre, WAIT_PREFLIGHT_S)
    if all(oks):
        try:
            ex = RcclExchange(rank, world, device)
        except Exception as e:                  # noqa: BLE001 -- any failure means "no RCCL on this rank"
            err = repr(e)
            complain("ncclCommInitRank", err)
        oks = agree("rccl_ok", ex is not None, WAIT_COMM_S)
    if all(oks):
        return ex
    if ex is not None:
        ex.close()
    fx = FileExchange(rank, world, device)
    fx.name = "file (RCCL unavailable on rank(s) %s%s)" % ([r for r, ok in enumerate(oks) if not ok], ": " + err if err else "")
    return fx


class RcclExchange:
    """The product exchange: RCCL inside the library (one ncclAllGather of 16 B x messages per rank per step)."""
    name = "rccl"

    def __init__(self, rank, world, device):
        uid = share_bytes("rccl_unique_id", lib.comm_unique_id() if rank == 0 else None, rank, 128)
        self.comm = lib.Comm(uid, world, rank, device=device)
        self.rank, self.world = self.comm.rank, self.comm.n_ranks          # as RCCL sees them
        if self.world != world or self.rank != rank:
            raise lib.AesGcmError(lib.ERCCL, "RCCL reports rank %d of %d, launcher said %d of %d" % (self.rank, self.world, rank, world))

    def allgather_dev(self, d_send, d_recv, bytes_per_rank, stream=None):
        self.comm.allgather_dev(d_send, d_recv, bytes_per_rank, stream)

    def allreduce(self, value, op="max"):
        return self.comm.allreduce(value, op)

    def barrier(self):
        self.comm.barrier()

    def close(self):
        self.comm.close()


class FileExchange:
    """Debug exchange through the host and a directory (several ranks on one GPU)."""
    name = "file"

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device, self.seq = rank, world, device, 0
        self.dir = rendezvous_dir()
        os.makedirs(self.dir, exist_ok=True)

    def _allgather_bytes(self, b):
        self.seq += 1
        _write_atomic(os.path.join(self.dir, "x%d_%d" % (self.seq, self.rank)), b)
        out = [_wait_read(os.path.join(self.dir, "x%d_%d" % (self.seq, r)), len(b), 300.0) for r in range(self.world)]
        if self.seq > 2:                     # everyone has passed seq-2 once it has written seq
            try:
                os.unlink(os.path.join(self.dir, "x%d_%d" % (self.seq - 2, self.rank)))
            except OSError:
                pass
        return out

    def allgather_dev(self, d_send, d_recv, bytes_per_rank, stream=None):
        L = lib.load()
        lib._chk(L.aesgcm_dev_sync(self.device))
        mine = bytearray(bytes_per_rank)
        lib._chk(L.aesgcm_dev_download(self.device, lib._Buf(mine, writable=True).addr, d_send, bytes_per_rank))
        allb = b"".join(self._allgather_bytes(bytes(mine)))
        lib._chk(L.aesgcm_dev_upload(self.device, d_recv, allb, len(allb)))

    def allreduce(self, value, op="max"):
        vals = [struct.unpack("<d", b)[0] for b in self._allgather_bytes(struct.pack("<d", value))]
        return max(vals) if op == "max" else min(vals) if op == "min" else sum(vals)

    def barrier(self):
        self._allgather_bytes(b"\x01")

    def close(self):
        pass
