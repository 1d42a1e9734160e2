"""A few steps of one update / time-step variant at C2 (run under rocprofv3 --kernel-trace; dev tool).
    python tools/variant_trace.py eks [time_step] [dense]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd import engine
update = sys.argv[1] if len(sys.argv) > 1 else "eks"
ts = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "none" else None
dense = len(sys.argv) > 3
p = n = 256; J = 65536
prob = bench.synthetic_problem(p, n)
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)); Gd = 0.01 * (B @ B.T / n + np.eye(n))
eng = engine.Engine(p, n, J, dtype="float32", seed=1)
eng.set_problem(prob["y"], Gd if dense else prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
U = torch.as_tensor(prob["ustar"], device="cuda", dtype=torch.float32) + torch.randn((p, J), device="cuda")
G = eng.forward_lineal(prob["A"], U); out = eng.empty(p)
t_last = 0.0
for i in range(6):
    prm = engine.step_params(update=update, time_step=ts, first_step=(i == 0), t_len=min(i, 1), t_last=t_last, delta_t=0.01, step_index=i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.step(prm, U, G, xi=None, out=out, recenter=(i == 0)); res = eng.result()
    print("%.3f ms" % (1e3 * (time.perf_counter() - t0))); t_last = res.t_new
