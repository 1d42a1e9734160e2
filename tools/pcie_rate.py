#!/usr/bin/env python3
"""dev tool: PCIe-inclusive rate of the drop-in path at C2 (host numpy in, host numpy out).

One step of ces_amd.calibrate.sampling.eks_update_aldi with host float64 arrays, as an unmodified
caller of the reference would use it: cast + H2D of U and G, the device step (device Philox noise),
D2H + cast of U_next.  Printed for DESIGN.md section 6; never bench.py's `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd.calibrate import sampling

p = n = 256; J = 65536
prob = bench.synthetic_problem(p, n)
rng = np.random.default_rng(0)
U = prob["ustar"] + rng.standard_normal((p, J))
G = prob["A"] @ U
eks = sampling(p=p, n_obs=n, J=J)
eks.mu, eks.sigma, eks.ustar = prob["mu"], prob["sigma"], prob["ustar"]
eks.engine_dtype, eks.noise = "float32", "device"
eks.Uall = [U, U]                                    # not the first step
eks.metrics = {k: [0.0] for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t")}
for _ in range(3):
    eks.eks_update_aldi(prob["y"], U, G, prob["Gamma"], 1)
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    Un = eks.eks_update_aldi(prob["y"], U, G, prob["Gamma"], 1)
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print("drop-in step with host float64 arrays in/out at C2 (fp32 engine): %.1f ms -> %.2f M particle-updates/s "
      "(min %.1f ms)" % (1e3 * t, J / t / 1e6, 1e3 * min(ts)))
t0 = time.perf_counter(); G2 = prob["A"] @ Un; tf = time.perf_counter() - t0
print("host forward map G = A U (numpy, %d threads): %.1f ms" % (os.cpu_count(), 1e3 * tf))
if os.environ.get("PCIE_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        eks.eks_update_aldi(prob["y"], U, G, prob["Gamma"], 1)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
