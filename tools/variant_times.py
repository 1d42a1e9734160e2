#!/usr/bin/env python3
"""dev tool: step time of every update rule / time-step rule / dense-vs-diagonal Gamma, Sigma at C2."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd import engine
p = n = 256; J = 65536
prob = bench.synthetic_problem(p, n)
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)); Gd = 0.01 * (B @ B.T / n + np.eye(n))
B = rng.standard_normal((p, p)); Sd = 100.0 * (B @ B.T / p + np.eye(p))
def run(update, ts, dense_g=False, dense_s=False):
    eng = engine.Engine(p, n, J, dtype="float32", seed=1)
    eng.set_problem(prob["y"], Gd if dense_g else prob["Gamma"], prob["mu"], Sd if dense_s else prob["sigma"], prob["ustar"])
    U = torch.as_tensor(prob["ustar"], device="cuda", dtype=torch.float32) + torch.randn((p, J), device="cuda")
    G = eng.forward_lineal(prob["A"], U); out = eng.empty(p)
    t_last, ts_ms = 0.0, []
    for i in range(8):
        prm = engine.step_params(update=update, time_step=ts, first_step=(i == 0), t_len=min(i, 1), t_last=t_last,
                                 delta_t=0.01, step_index=i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.step(prm, U, G, xi=None, out=out, recenter=(i == 0)); res = eng.result()
        ts_ms.append(1e3 * (time.perf_counter() - t0)); t_last = res.t_new
    return min(ts_ms[2:])
for update in ("aldi", "eks", "aldi_constant"):
    for ts in ((None, "spectral", "constant", "mix") if update != "aldi_constant" else (None,)):
        print("%-14s time_step=%-9s %.3f ms" % (update, ts, run(update, ts)))
print("aldi dense Gamma          %.3f ms" % run("aldi", None, dense_g=True))
print("aldi dense Sigma          %.3f ms" % run("aldi", None, dense_s=True))
print("eks  dense Gamma + Sigma  %.3f ms" % run("eks", None, True, True))
