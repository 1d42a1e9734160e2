"""Per-iteration phase times of sampling.run(trace=False) with a host forward map at C2 (dev tool)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_problem, limit_host_threads
from ces_amd import engine as _E0
nthr = limit_host_threads(reserve=_E0.Engine.copy_threads)
if os.environ.get("PROBE_THREADS"):
    nthr = int(os.environ["PROBE_THREADS"])
    from threadpoolctl import threadpool_limits
    threadpool_limits(nthr)
    torch.set_num_threads(nthr)
def cpu_stat():
    try:
        return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception as ex:
        return {"error": repr(ex)}
print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None, "threads", nthr)
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
FWD_OUT = os.environ.get("PROBE_FWD_OUT") == "1"          # the forward map writes into one reused buffer (no fresh 134 MB array)
NO_DISCARD = os.environ.get("PROBE_NO_DISCARD") == "1"    # nothing is handed back to the pool (no recycling, no munmap on the helper)
if NO_DISCARD:
    E.Engine.discard_host = lambda self, *a: None
_buf = [None]
def g_ens(theta, m):
    t0 = time.perf_counter()
    if FWD_OUT:
        if _buf[0] is None: _buf[0] = np.empty((n, theta.shape[1]))
        g = np.matmul(prob["A"], theta, out=_buf[0])
    else:
        g = prob["A"] @ theta
    log.append(("forward", 1e3 * (time.perf_counter() - t0))); return g
eks = sampling(p=p, n_obs=n, J=J)
eks.mu, eks.sigma, eks.ustar = prob["mu"], prob["sigma"], prob["ustar"]
eks.engine_dtype, eks.noise, eks.device, eks.device_loop = "float32", "device", 0, False
eks.G_ens = g_ens
eks.T = 2
eks.run(prob["y"], U0, host_lineal(), prob["Gamma"], None, trace=False, t_tol=1e30)
eks.T = 30
del log[:]
log.append(("iter", 0.0))
if os.environ.get("PROBE_COPY_THREADS"):
    E.Engine.copy_threads = int(os.environ["PROBE_COPY_THREADS"])
st0 = cpu_stat()
t0 = time.perf_counter()
eks.run(prob["y"], eks.Ustar, host_lineal(), prob["Gamma"], None, trace=False, t_tol=1e30)
el = time.perf_counter() - t0
st1 = cpu_stat()
print("cpu.stat delta:", {k: st1[k] - st0[k] for k in st1 if k in st0 and isinstance(st1[k], int)}, "wall_us", int(el * 1e6))
print("%.2f ms per iteration" % (1e3 * el / 30))
line = []
for name, ms in log:
    if name == "forward" and line:
        print("  ".join(line)); line = []
    line.append("%s %.2f" % (name, ms))
print("  ".join(line))
pool = eks._engine.__dict__.get("_out_pool")
print("recycled", getattr(pool, "recycled", None), "ready", {k: len(v) for k, v in pool.ready.items()}, "pending", pool.pending)
import collections
by = collections.defaultdict(list)
for name, ms in log:
    by[name].append(ms)
for name, v in by.items():
    v = np.array(v)
    print("%-14s n=%3d  p50 %.2f  p95 %.2f  max %.2f  sum/iter %.2f" % (name, len(v), np.median(v), np.percentile(v, 95), v.max(), v.sum() / 30))
