import os, sys
sys.path.insert(0, os.getcwd())
os.environ["AESGCM_TW"] = "1"
os.environ["AESGCM_BODY_MIN"] = str(1 << 60)
import aesgcm_amd
from aesgcm_amd import lib as hip
from oracle import oracle as orc
key, iv = bytes(range(32)), bytes(12)
ctx, f = hip.Context(key), orc.Fast(key)
for rows in (1024, 1025, 8192, 8193, 16384, 16385, 17000, 20000, 32768, 40000, 65535, 65536, 65537, 70000):
    n = rows * 1024
    d_in = hip.DeviceBuffer(n + 32); d_in.fill_splitmix64(4400 + rows, 0, nbytes=n)
    pt = bytes(d_in.download(n))
    want = f.encrypt(iv, b"", pt)
    d_out = hip.DeviceBuffer(n + 32)
    tags = [ctx.encrypt_dev(iv, d_in.ptr, n, d_out.ptr) for _ in range(3)]
    print(rows, [t == want[1] for t in tags], flush=True)
