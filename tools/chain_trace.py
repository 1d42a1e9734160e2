"""Kernel time line of the chained device-resident loop (run under rocprofv3 --kernel-trace; dev tool)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ces_amd import engine
from ces_amd.dist import ShardedSampler
from ces_amd.utils import lineal
from bench import synthetic_problem
p = n = 256; J = 65536
prob = synthetic_problem(p, n)
eng = engine.Engine(p, n, J, dtype="float32")
rng = np.random.default_rng(3)
U0 = prob["ustar"] + rng.standard_normal((p, J))
model = lineal(prob["A"])
for T in (40, 40):
    smp = ShardedSampler(eng, p, n, J); smp.T = T
    torch.cuda.synchronize(); t0 = time.perf_counter()
    smp.run(prob["y"], U0, model, prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"], t_tol=1e30); torch.cuda.synchronize()
    print("sampler.run per step ms", (time.perf_counter() - t0) / T * 1e3)
