#!/usr/bin/env python3
"""dev tool: host time of the calls one pipelined step makes (C4 shape by default): where the driving thread's
65 us per step go -- Python / ctypes plumbing or the HIP launches inside the library."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd import engine
from ces_amd.dist import ShardedUpdate
p, n, J = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 50, 8192)))
prob = bench.synthetic_problem(p, n)
eng = engine.Engine(p, n, J, dtype="float32", device=0, seed=1)
eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
sh = ShardedUpdate(eng)
dev = torch.device("cuda", 0)
U = torch.as_tensor(prob["ustar"], device=dev, dtype=torch.float32) + torch.randn((p, J), device=dev, dtype=torch.float32)
G = eng.forward_lineal(prob["A"], U)
out = eng.empty(p)
prm0 = engine.step_params(update="aldi")
sh.begin(prm0, U, G, recenter=True, noise_step=0)
acc = {}
def T(name, f, *a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e6); return r
t_last = 0.0
for i in range(3000):
    prm = T("step_params", engine.step_params, update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i)
    T("apply", eng.apply, prm, sh._mom, U, G, xi=None, out=out)
    T("prefetch_noise", eng.prefetch_noise, i + 1)
    mom = sh._moment_buffer()
    T("moments_uu_chol", eng.moments_uu_chol, prm0, U, G, out=mom)
    T("moments_rest", eng.moments_rest, U, G, mom)
    sh._mom = mom
    res = T("result", eng.result)
    t_last = res.t_new if i % 1000 else 0.0
    T("stream()", eng._stream)
for k, v in acc.items():
    v = np.array(v[500:])
    print("%-16s median %6.1f us  mean %6.1f" % (k, np.median(v), v.mean()))
