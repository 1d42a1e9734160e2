import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ces_amd.engine import Engine
P = Engine._HostOutPool
orig = P._fresh.__func__
def traced(cls, shape, threads=8):
    t0 = time.perf_counter(); a = orig(cls, shape, threads); print("   [worker] fresh %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True); return a
P._fresh = classmethod(traced)
pool = P()
for r in range(5):
    t0 = time.perf_counter(); a = pool.get((256, 65536)); t1 = time.perf_counter()
    print("get %.2f ms, qsize %d" % ((t1 - t0) * 1e3, pool.q.qsize()), flush=True)
    a[:] = 1.0
    time.sleep(0.03)
    t0 = time.perf_counter(); del a; print("del %.2f ms" % ((time.perf_counter() - t0) * 1e3))
