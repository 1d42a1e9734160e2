"""dev tool: phase timing of the host-array drop-in path (to_device x2, step+result, to_host)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd import engine
p = n = 256; J = 65536
prob = bench.synthetic_problem(p, n)
rng = np.random.default_rng(0)
U = prob["ustar"] + rng.standard_normal((p, J)); G = prob["A"] @ U
eng = engine.Engine(p, n, J, dtype="float32", seed=1)
eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
prm = engine.step_params(update="aldi", first_step=False, t_len=1, t_last=0.1, step_index=1)
keep = []
torch.set_num_threads(int(os.environ.get("NT", "16")))
print("torch threads", torch.get_num_threads())
for it in range(12):
    t0 = time.perf_counter(); Ud = eng.to_device(U, p, "U"); t1 = time.perf_counter()
    Gd = eng.to_device(G, n, "G"); t2 = time.perf_counter()
    out = eng.step(prm, Ud, Gd, xi=None); res = eng.result(); torch.cuda.synchronize(); t3 = time.perf_counter()
    h = eng.to_host(out); t4 = time.perf_counter()
    keep.append(h)
    print("it %2d  U %.1f  G %.1f  step %.1f  to_host %.1f ms" % (it, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)))
