"""CTR-only keystream rate of the T-table kernel (k_main<NR, MODE_KS>: no input, no GHASH, 16 B written per block) --
the figure the bitsliced microbenchmark (profiles/microbench/bs_ctr.hip) is put beside."""
import sys
import time
sys.path.insert(0, ".")
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402

n = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096) << 20
buf = lib.DeviceBuffer(n)
for kb in (16, 32):
    ctx = lib.Context(bytes(range(kb)))
    iv = bytes(range(12))
    ctx.keystream_dev(iv, 0, n // 16, buf.ptr); lib.dev_sync(0)
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.keystream_dev(iv, 0, n // 16, buf.ptr); lib.dev_sync(0)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print("k_main<%d,KS> T-table AES-%d CTR keystream, %d MiB: %.3f ms  %.1f GB/s (%.1f GiB/s)" % (kb // 4 + 6, 8 * kb, n >> 20, best * 1e3, n / best / 1e9, n / best / 2**30))
