import sys, time, os
sys.path.insert(0, "/root/repo")
import aesgcm_amd
from aesgcm_amd import lib
lib.Context(bytes(32)).close()
ts=[]
for i in range(20):
    t0=time.perf_counter(); c=lib.Context(bytes([i])*32); t1=time.perf_counter(); c.close(); t2=time.perf_counter()
    ts.append((t1-t0, t2-t1))
ts.sort()
print("ctx create median %.1f us, destroy median %.1f us" % (ts[10][0]*1e6, sorted(x[1] for x in ts)[10]*1e6))
