import os, sys
sys.path.insert(0, os.getcwd())
import aesgcm_amd
from aesgcm_amd import lib
os.environ["AESGCM_PKT_SHAPE"] = "w"
ctx = lib.Context(bytes(range(16)))
for n, pkt in ((1, 16), (300, 4096)):
    d_ivs, d_in, d_out, d_tags = lib.DeviceBuffer(max(12 * n, 16)), lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(16 * n)
    d_in.fill_splitmix64(3)
    print("launch", n, pkt, flush=True)
    ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=pkt)
    lib.dev_sync()
    print("done", n, pkt, bytes(d_tags.download())[:16].hex(), flush=True)
