import os, sys, time
sys.path.insert(0, os.getcwd())
import aesgcm_amd
from aesgcm_amd import lib
from oracle import oracle as O
key = bytes(range(16)); ctx = lib.Context(key); f = O.Fast(key)
def up(b):
    d = lib.DeviceBuffer(max(len(b), 16)); d.upload(b); return d
for n, pkt, al in ((1, 16, 0), (1, 4096, 0), (3, 4096, 20), (40, 4096, 20), (300, 4096, 20)):
    ivs, aad, pt = bytes(O.fill_splitmix64(12 * n, 31)), bytes(O.fill_splitmix64(al * n, 32)), bytes(O.fill_splitmix64(pkt * n, 33))
    d_ivs, d_aad, d_in = up(ivs), up(aad), up(pt)
    d_out, d_tags = lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(16 * n)
    print("launch", n, pkt, al, flush=True)
    ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=pkt, d_aad=d_aad.ptr if al else None, aad_len=al)
    lib.dev_sync()
    ct, tags = bytes(d_out.download()), bytes(d_tags.download())
    ok = all((ct[pkt * p:pkt * (p + 1)], tags[16 * p:16 * p + 16]) == f.encrypt(ivs[12 * p:12 * p + 12], aad[al * p:al * (p + 1)], pt[pkt * p:pkt * (p + 1)]) for p in range(n))
    print("done", n, pkt, al, ok, flush=True)
