"""dev tool: phase times of the column-block-pipelined host loop (sampling._run_host_pipelined) at C2."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_problem, limit_host_threads
from ces_amd import engine as E
nthr = limit_host_threads(reserve=E.Engine.copy_threads)
p = n = 256; J = 65536
prob = synthetic_problem(p, n)
A = prob["A"]
rng = np.random.default_rng(3)
U = prob["ustar"] + rng.standard_normal((p, J))
eng = E.Engine(p, n, J, dtype="float32")
pin_u = eng._pinned("hp_u", (p, J)); pin_g = eng._pinned("hp_g0", (n, J // 4))
Ud = eng.to_device(U, p, "U"); Gd = eng.empty(n)
torch.cuda.synchronize()
def t(f, reps=5):
    v = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); v.append(1e3 * (time.perf_counter() - t0))
    return "%.2f (min %.2f)" % (np.median(v), min(v))
a, b = 0, J // 4
Uh = np.empty((p, J))
print("threads blas", nthr, "copy", eng.copy_threads)
print("forward full           ", t(lambda: A @ U))
print("forward block (view)   ", t(lambda: A @ U[:, a:b]))
Uc = np.ascontiguousarray(U[:, a:b])
print("forward block (contig) ", t(lambda: A @ Uc))
Gc = A @ Uc
def widen():
    with eng._HostThreads(eng.copy_threads):
        torch.from_numpy(Uh)[:, a:b].copy_(pin_u[:, a:b])
print("widen block strided    ", t(widen))
tmp = torch.empty((p, b - a), dtype=torch.float64)
def widen2():
    with eng._HostThreads(eng.copy_threads):
        tmp.copy_(pin_u[:, a:b])
print("widen block -> contig  ", t(widen2))
def cast():
    with eng._HostThreads(eng.copy_threads):
        pin_g.copy_(torch.from_numpy(Gc))
print("cast G block           ", t(cast))
def h2d():
    Gd[:, a:b].copy_(pin_g, non_blocking=True); torch.cuda.synchronize()
print("H2D G block (2D)       ", t(h2d))
def d2h():
    pin_u[:, a:b].copy_(Ud[:, a:b], non_blocking=True); torch.cuda.synchronize()
print("D2H U block (2D)       ", t(d2h))
def d2h_full():
    pin_u.copy_(Ud, non_blocking=True); torch.cuda.synchronize()
print("D2H U full             ", t(d2h_full))
# row blocks instead of column blocks
r0, r1 = 0, p // 4
def d2h_rows():
    pin_u[r0:r1].copy_(Ud[r0:r1], non_blocking=True); torch.cuda.synchronize()
print("D2H U row block        ", t(d2h_rows))
def widen_rows():
    with eng._HostThreads(eng.copy_threads):
        torch.from_numpy(Uh)[r0:r1].copy_(pin_u[r0:r1])
print("widen row block        ", t(widen_rows))

# the loop itself, phase by phase
from ces_amd.calibrate import sampling
class host_lineal:
    type, model_name, n_obs = "map", "lineal", n
    def __call__(self, theta): return A @ theta
eks = sampling(p=p, n_obs=n, J=J)
eks.mu, eks.sigma, eks.ustar = prob["mu"], prob["sigma"], prob["ustar"]
eks.engine_dtype, eks.noise, eks.device = "float32", "device", 0
eks.G_ens = lambda theta, m: A @ theta
for T in (3, 20):
    eks.T = T
    t0 = time.perf_counter()
    eks.run(prob["y"], U, host_lineal(), prob["Gamma"], None, trace=False, t_tol=1e30)
    el = time.perf_counter() - t0
    print("T=%d: %.2f ms per iteration (pipelined host loop, incl. set-up and the final forward evaluation); waits (ms per iteration):" % (T, 1e3 * el / T),
          {k: round(1e3 * v / T, 2) for k, v in eks._pipe_times.items()})
eks.host_pipeline = False
eks.T = 20
t0 = time.perf_counter()
eks.run(prob["y"], U, host_lineal(), prob["Gamma"], None, trace=False, t_tol=1e30)
print("T=20: %.2f ms per iteration (plain host loop)" % (1e3 * (time.perf_counter() - t0) / 20))
