"""CPU, world_size 2, gloo: the N > 1 orchestration -- shard planning, per-rank weighted partials, ONE
16-byte-per-rank all-gather, fold, finalize -- with the oracle standing in for the GPU compute (tests may
use the oracle; the product path on a GPU box runs the same plan through libaesgcm_hip.so, see bench.py)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import sharding
    from oracle import oracle as O
    from util import splitmix_bytes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    key, iv, aad = splitmix_bytes(0x7001, 32), splitmix_bytes(0x7002, 12), splitmix_bytes(0x7003, case["aad"])
    n = case["len"]
    pt = splitmix_bytes(0x7004, n)
    f = O.Fast(key)
    h = f.h
    first, end, ln = sharding.shard_bounds(n, world, rank)
    n_blocks = (n + 15) // 16
    # this rank's shard: CTR at its counter offset, polynomial over (AAD on rank 0) + its ciphertext blocks
    ks = f.keystream(iv, first, end - first)
    ct = bytes(a ^ b for a, b in zip(pt[16 * first:16 * first + ln], ks))
    seq = (aad + bytes(-len(aad) % 16) if rank == 0 else b"") + ct
    w = O.gfmul(O.gfpow(h, n_blocks - end), f.ghash_poly(seq)) if seq else bytes(16)
    mine = torch.tensor(list(w), dtype=torch.uint8)
    gathered = [torch.zeros(16, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gathered, mine)                      # the one collective of the data path
    fold = bytes(16)
    for g in gathered:
        fold = bytes(a ^ b for a, b in zip(fold, bytes(g.tolist())))
    lb = (8 * len(aad)).to_bytes(8, "big") + (8 * n).to_bytes(8, "big")
    y = O.gfmul(h, bytes(a ^ b for a, b in zip(O.gfmul(h, fold), lb)))
    ej0 = f.encrypt_block(iv + b"\x00\x00\x00\x01")
    tag = bytes(a ^ b for a, b in zip(y, ej0))
    want_ct, want_tag = f.encrypt(iv, aad, pt)
    ok = (tag == want_tag) and (ct == want_ct[16 * first:16 * first + ln])
    q.put((rank, ok, tag.hex()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [dict(aad=37, len=203 * 16 + 5), dict(aad=0, len=16 * 1001), dict(aad=20, len=7)])
def test_two_rank_sharded_message_gloo(case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({t for _, _, t in res}) == 1          # every rank derives the same tag


def test_plan_job_shapes():
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import sharding
    GiB = 1 << 30
    for n in (1, 2, 4, 8):
        per_rank = [sharding.plan_job(n, 16 * GiB, r) for r in range(n)]
        n_msgs = {1: 1, 2: 1, 4: 2, 8: 4}[n]
        for r, plan in enumerate(per_rank):
            assert len(plan) == n_msgs
            assert sum(m["len"] for m in plan) == 16 * GiB
            for m in plan:
                assert m["total"] <= sharding.MAX_MESSAGE
                assert m["total"] == (16 * GiB if n == 1 else 32 * GiB)
        # shards of one message tile it exactly, in rank order, and the plaintext stream is contiguous
        for mi in range(n_msgs):
            blocks = 0
            for r in range(n):
                m = per_rank[r][mi]
                assert m["first_block"] == blocks
                assert m["stream_word"] == (mi * m["total"] + 16 * blocks) // 8
                blocks += m["len"] // 16
            assert blocks * 16 == per_rank[0][mi]["total"]
    assert sharding.shard_bounds(203 * 16 + 5, 8, 7) == (179, 204, 389)
    assert sharding.shard_bounds(7, 8, 0) == (0, 1, 7) and sharding.shard_bounds(7, 8, 5) == (1, 1, 0)
    with pytest.raises(ValueError):
        sharding.plan_job(1, 1 << 37, 0)


def _file_worker(rank, world, rdzv, q):
    sys.path.insert(0, ROOT)
    os.environ["AESGCM_RDZV_DIR"] = rdzv
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import comm
    uid = comm.share_bytes("rccl_unique_id", bytes(range(128)) if rank == 0 else None, rank, 128)
    ex = comm.FileExchange(rank, world, 0)
    got = []
    for k in range(5):                                   # several rounds: sequence numbers and file reuse
        got.append(ex._allgather_bytes(bytes([rank, k])))
    mx = ex.allreduce(float(10 + rank), "max")
    mn = ex.allreduce(float(10 + rank), "min")
    ex.barrier()
    comm.finish(rank, world)
    q.put((rank, uid == bytes(range(128)), got, mx, mn))


def test_launcher_plumbing_without_torch_two_ranks(tmp_path):
    """the torch-free rank plumbing bench.py uses for N > 1: unique-id hand-off through a file, the debug file exchange
    (all-gather / all-reduce / barrier) and the goodbye protocol that removes the rendezvous directory"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdzv = str(tmp_path / "rdzv")
    procs = [ctx.Process(target=_file_worker, args=(r, 2, rdzv, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, uid_ok, got, mx, mn in res:
        assert uid_ok and mx == 11.0 and mn == 10.0
        assert got == [[bytes([0, k]), bytes([1, k])] for k in range(5)]
    assert not os.path.exists(rdzv)                      # rank 0 removed it after every goodbye


def test_tweak_iv_refuses_to_wrap():
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import sharding
    iv = bytes(11) + b"\xfe"
    assert sharding.tweak_iv(iv, 1)[11] == 0xFF
    with pytest.raises(ValueError):
        sharding.tweak_iv(iv, 2)
    with pytest.raises(ValueError):
        sharding.plan_job(8, 1 << 30, 0, msg_bytes=16 * 8 * 2)       # 2^22 messages: more than one IV byte can number
