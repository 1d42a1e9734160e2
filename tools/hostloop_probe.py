"""Per-iteration phase times of sampling.run(trace=False) with a host forward map at C2 (dev tool)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_problem, limit_host_threads
limit_host_threads()
from ces_amd.calibrate import sampling
from ces_amd import engine as E
p = n = 256; J = 65536
prob = synthetic_problem(p, n)
rng = np.random.default_rng(3)
U0 = prob["ustar"] + rng.standard_normal((p, J))
class host_lineal:
    type, model_name, n_obs = "map", "lineal", n
    def __call__(self, theta): return prob["A"] @ theta
log = []
def timed(name, f):
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); log.append((name, 1e3 * (time.perf_counter() - t0))); return r
    return g
for name in ("to_device", "to_host", "step", "result", "discard_host"):
    setattr(E.Engine, name, timed(name, getattr(E.Engine, name)))
def g_ens(theta, m):
    t0 = time.perf_counter(); g = prob["A"] @ theta; log.append(("forward", 1e3 * (time.perf_counter() - t0))); return g
eks = sampling(p=p, n_obs=n, J=J)
eks.mu, eks.sigma, eks.ustar = prob["mu"], prob["sigma"], prob["ustar"]
eks.engine_dtype, eks.noise, eks.device, eks.device_loop = "float32", "device", 0, False
eks.G_ens = g_ens
eks.T = 2
eks.run(prob["y"], U0, host_lineal(), prob["Gamma"], None, trace=False, t_tol=1e30)
eks.T = 10
del log[:]
log.append(("iter", 0.0))
t0 = time.perf_counter()
eks.run(prob["y"], eks.Ustar, host_lineal(), prob["Gamma"], None, trace=False, t_tol=1e30)
el = time.perf_counter() - t0
print("%.2f ms per iteration" % (1e3 * el / 10))
line = []
for name, ms in log:
    if name == "forward" and line:
        print("  ".join(line)); line = []
    line.append("%s %.2f" % (name, ms))
print("  ".join(line))
pool = eks._engine.__dict__.get("_out_pool")
print("recycled", getattr(pool, "recycled", None), "qsize", pool.q.qsize(), "pending", pool.pending)
