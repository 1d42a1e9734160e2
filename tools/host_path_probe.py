"""Phase timing of the host-array drop-in update (float64 numpy in / out) at C2 (dev tool)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ces_amd import engine
from bench import synthetic_problem
p = n = 256; J = 65536
prob = synthetic_problem(p, n)
eng = engine.Engine(p, n, J, dtype="float32")
eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
rng = np.random.default_rng(3)
U0 = prob["ustar"] + rng.standard_normal((p, J)); G0 = prob["A"] @ U0
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
def T(f, reps=6):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return "first %.1f ms, rest median %.2f ms, min %.2f" % (ts[0], float(np.median(ts[1:])), min(ts[1:]))
print("to_device(U)      ", T(lambda: eng.to_device(U0, tag="U")))
print("to_device(G)      ", T(lambda: eng.to_device(G0, tag="G")))
Ud, Gd = eng.to_device(U0, tag="U"), eng.to_device(G0, tag="G")
prm = engine.step_params(update="aldi")
def st():
    o = eng.step(prm, Ud, Gd, recenter=True); eng.result(); return o
print("step+result (dev) ", T(st))
o = st()
print("to_host           ", T(lambda: eng.to_host(o)))
pin = eng._pinned("out", o.shape)
print("  pin.copy_(dev)  ", T(lambda: pin.copy_(o)))
print("  pin->f64 (torch)", T(lambda: pin.to(torch.float64)))
for th in (4, 8, 16, 32):
    torch.set_num_threads(th)
    print("  pin->f64 threads=%d" % th, T(lambda: pin.to(torch.float64)))
    print("  f64->pin threads=%d" % th, T(lambda: pin.copy_(torch.from_numpy(U0))))
print("  np astype f64   ", T(lambda: pin.numpy().astype(np.float64)))
print("  np empty+copyto ", T(lambda: np.copyto(np.empty((p, J)), pin.numpy())))
def full():
    return eng.to_host(eng.step(prm, U0, G0, recenter=True)) if eng.result is None else None
def full2():
    out = eng.step(prm, U0, G0, recenter=True); eng.result(); return eng.to_host(out)
print("full host call    ", T(full2))
print("--- to_host phases (back-to-back calls)")
pool = eng.__dict__.setdefault("_out_pool", eng._HostOutPool())
for r in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pin.copy_(o); t1 = time.perf_counter()
    out = pool.get(tuple(o.shape)); t2 = time.perf_counter()
    with eng._HostThreads(eng.copy_threads):
        torch.from_numpy(out).copy_(pin)
    t3 = time.perf_counter()
    print("  D2H %.2f  pool.get %.2f  widen %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print("--- with 20 ms between calls")
for r in range(4):
    time.sleep(0.02)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.to_host(o); t1 = time.perf_counter()
    print("  to_host %.2f ms" % ((t1 - t0) * 1e3))
