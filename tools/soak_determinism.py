#!/usr/bin/env python3
"""dev tool: race screen.  The same step (same inputs, same Philox step index) is repeated many
times; every output must be bit-identical to the first (the kernels use no float atomics, so any
difference is a synchronisation bug, e.g. in the LDS-DMA ring of kernels_update2.hip)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd import engine

def screen(p, n, J, update, reps, dtype="float32"):
    prob = bench.synthetic_problem(p, n)
    eng = engine.Engine(p, n, J, dtype=dtype, seed=7)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    g = torch.Generator(device="cuda").manual_seed(1)
    U = torch.as_tensor(prob["ustar"], device="cuda", dtype=eng.torch_dtype) + torch.randn((p, J), generator=g, device="cuda", dtype=eng.torch_dtype)
    G = eng.forward_lineal(prob["A"], U)
    prm = engine.step_params(update=update, first_step=False, t_len=1, t_last=0.1, step_index=3)
    ref, ref_res, bad = None, None, 0
    for r in range(reps):
        out = eng.step(prm, U, G, xi=None, recenter=True)
        res = eng.result()
        key = (res.hk, res.bias_data, res.self_bias_data)
        if ref is None:
            ref, ref_res = out.clone(), key
        elif not torch.equal(out, ref) or key != ref_res:
            bad += 1
    print("%-14s %s p=%d n=%d J=%d: %d repeats, %d differ" % (update, dtype, p, n, J, reps, bad))
    return bad

bad = 0
bad += screen(256, 256, 65536, "aldi", 400)
bad += screen(256, 256, 65536, "aldi_constant", 100)
bad += screen(256, 256, 65536, "eks", 60)
bad += screen(300, 40, 50004, "aldi", 200)
bad += screen(96, 80, 5000, "aldi", 400)
bad += screen(512, 512, 8192, "aldi", 60)
bad += screen(256, 256, 16384, "aldi", 60, "float64")
print("TOTAL differing repeats:", bad)
sys.exit(1 if bad else 0)
