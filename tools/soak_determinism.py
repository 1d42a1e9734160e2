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

def chain(p, n, J, nsteps, pipelined, dtype="float32"):
    """A chained run through ShardedUpdate (the drivers' loop): the output of step i is the input of step i + 1, G = A U
    on the device.  pipelined: begin(i+1) before result(i) -- the path with the deferred publication, the bound
    hand-over events and the noise lookahead all in play."""
    from ces_amd.dist import ShardedUpdate
    prob = bench.synthetic_problem(p, n)
    eng = engine.Engine(p, n, J, dtype=dtype, seed=11)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    sh = ShardedUpdate(eng)
    g = torch.Generator(device="cuda").manual_seed(5)
    U = torch.as_tensor(prob["ustar"], device="cuda", dtype=eng.torch_dtype) + torch.randn((p, J), generator=g, device="cuda", dtype=eng.torch_dtype)
    A = torch.as_tensor(prob["A"], device="cuda", dtype=eng.torch_dtype)
    bufs = [eng.empty(p), eng.empty(p)]
    Gs = [eng.forward_lineal(A, U), None]
    t_last, hks = 0.0, []
    def prm(i, t_last):
        return engine.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i)
    sh.begin(prm(0, 0.0), U, Gs[0], recenter=True, noise_step=0)
    for i in range(nsteps):
        out = sh.finish(prm(i, t_last), U, Gs[i % 2], xi=None, out=bufs[i % 2])
        if i + 1 < nsteps:
            Gs[(i + 1) % 2] = eng.forward_lineal(A, out)
            if pipelined:
                sh.begin(prm(i + 1, 0.0), out, Gs[(i + 1) % 2], noise_step=i + 1)
        res = sh.result()
        if i + 1 < nsteps and not pipelined:
            sh.begin(prm(i + 1, 0.0), out, Gs[(i + 1) % 2], noise_step=i + 1)
        t_last = res.t_new
        hks.append((res.hk, res.bias_data, res.self_bias_data, res.bias))
        U = out
    return U.clone(), hks

def chain_screen(p, n, J, nsteps, dtype="float32"):
    a, ha = chain(p, n, J, nsteps, True, dtype)
    b, hb = chain(p, n, J, nsteps, True, dtype)
    c, hc = chain(p, n, J, nsteps, False, dtype)
    # the side stream joined with the event instead of the polled word (round 3): the same arithmetic, so any
    # difference is a stale read of what that stream produced
    os.environ["CESX_POLL_JOIN"] = "0"
    try:
        d, hd = chain(p, n, J, nsteps, True, dtype)
    finally:
        del os.environ["CESX_POLL_JOIN"]
    bad = (int(not torch.equal(a, b)) + int(ha != hb) + int(not torch.equal(a, c)) + int(ha != hc)
           + int(not torch.equal(a, d)) + int(ha != hd))
    print("chain %s p=%d n=%d J=%d, %d steps: pipelined twice + step-by-step + event-joined, %d mismatches (t_end %.4f)" %
          (dtype, p, n, J, nsteps, bad, sum(h[0] for h in ha)))
    return bad

bad = 0
bad += chain_screen(256, 256, 65536, 1500)
bad += chain_screen(96, 80, 5000, 1500)
bad += chain_screen(256, 256, 8192, 400, "float64")
bad += screen(256, 256, 65536, "aldi", 400)
bad += screen(256, 256, 65536, "aldi_constant", 100)
bad += screen(256, 256, 65536, "eks", 60)
bad += screen(300, 40, 50004, "aldi", 200)
bad += screen(96, 80, 5000, "aldi", 400)
bad += screen(512, 512, 8192, "aldi", 60)
bad += screen(256, 256, 16384, "aldi", 60, "float64")
print("TOTAL differing repeats:", bad)
sys.exit(1 if bad else 0)
