"""Where does a device-resident chained step spend its time?  (dev tool)"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ces_amd import engine
from ces_amd.dist import ShardedSampler, ShardedUpdate
from ces_amd.utils import lineal
from bench import synthetic_problem

p = n = 256; J = 65536
prob = synthetic_problem(p, n)
eng = engine.Engine(p, n, J, dtype="float32")
eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
rng = np.random.default_rng(3)
U0 = prob["ustar"] + rng.standard_normal((p, J))
U = eng.to_device(U0)
model = lineal(prob["A"])
def T(f, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("forward_device ms", T(lambda: model.forward_device(eng, U)))
A_d = torch.as_tensor(prob["A"], device="cuda", dtype=torch.float32)
print("forward_lineal (A on device) ms", T(lambda: eng.forward_lineal(A_d, U)))
G = eng.forward_lineal(A_d, U)
sh = ShardedUpdate(eng)
prm = engine.step_params(update="aldi")
def step():
    sh.begin(prm, U, G, recenter=False); out = sh.finish(prm, U, G); eng.result(); return out
sh.begin(prm, U, G, recenter=True); sh.finish(prm, U, G); eng.result()
print("step (begin+finish+result) ms", T(step))
def step2():
    out = eng.step(prm, U, G, recenter=False); eng.result(); return out
print("eng.step ms", T(step2))
smp = ShardedSampler(eng, p, n, J); smp.T = 20
t0 = time.perf_counter(); smp.run(prob["y"], U, model, prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"], t_tol=1e30); torch.cuda.synchronize()
print("sampler.run per step ms", (time.perf_counter() - t0) / 20 * 1e3, smp.metrics["t"][-3:])
smp = ShardedSampler(eng, p, n, J); smp.T = 20
t0 = time.perf_counter(); smp.run(prob["y"], U0, model, prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"], t_tol=1e30); torch.cuda.synchronize()
print("sampler.run(host U0) per step ms", (time.perf_counter() - t0) / 20 * 1e3)
# host arrays
t = T(lambda: eng.to_device(U0, tag="U"), 5); print("to_device U0 f64->f32 ms", t)
o = eng.empty(p)
t = T(lambda: eng.to_host(o), 5); print("to_host ms", t)
