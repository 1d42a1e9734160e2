#!/usr/bin/env python3
"""dev tool: long chains of the rules that carry a warm-started SPD inverse (`eks`, time_step='constant' / 'mix') -- pipelined
twice + step by step must be bit-identical (the closing residual's ticket and agent-scope partials, the two-buffer inverse,
the sweep count sized from the last step's residual: the flows read a step's result before they enqueue the next apply, so
they size alike)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd import engine
from ces_amd.dist import ShardedUpdate


def chain(p, n, J, nsteps, pipelined, update, kw, dtype="float32", dense=False):
    prob = bench.synthetic_problem(p, n, dense_gamma=dense, dense_sigma=dense)
    eng = engine.Engine(p, n, J, dtype=dtype, seed=11)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    sh = ShardedUpdate(eng)
    g = torch.Generator(device="cuda").manual_seed(5)
    U = torch.as_tensor(prob["ustar"], device="cuda", dtype=eng.torch_dtype) + torch.randn((p, J), generator=g, device="cuda", dtype=eng.torch_dtype)
    A = torch.as_tensor(prob["A"], device="cuda", dtype=eng.torch_dtype)
    bufs = [eng.empty(p), eng.empty(p)]
    Gs = [eng.forward_lineal(A, U), None]
    t_last, hks, warm = 0.0, [], 0

    def prm(i, t_last):
        return engine.step_params(update=update, first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i, **kw)
    prm0 = engine.step_params(update=update)
    sh.begin(prm0, U, Gs[0], recenter=True, noise_step=0)
    for i in range(nsteps):
        out = sh.finish(prm(i, t_last), U, Gs[i % 2], xi=None, out=bufs[i % 2])
        if i + 1 < nsteps:
            Gs[(i + 1) % 2] = eng.forward_lineal(A, out)
            if pipelined:
                sh.begin(prm0, out, Gs[(i + 1) % 2], noise_step=i + 1)
        res = sh.result()
        if i + 1 < nsteps and not pipelined:
            sh.begin(prm0, out, Gs[(i + 1) % 2], noise_step=i + 1)
        t_last = res.t_new
        hks.append((res.hk, res.bias_data, res.self_bias_data, res.bias))
        U = out
        if i % 100 == 99:
            warm += eng.warm_inverse()
    return U.clone(), hks, warm


def screen(p, n, J, nsteps, update, kw, dtype="float32", dense=False):
    a, ha, wa = chain(p, n, J, nsteps, True, update, kw, dtype, dense)
    b, hb, _ = chain(p, n, J, nsteps, True, update, kw, dtype, dense)
    c, hc, _ = chain(p, n, J, nsteps, False, update, kw, dtype, dense)
    bad = int(not torch.equal(a, b)) + int(ha != hb) + int(not torch.equal(a, c)) + int(ha != hc)
    fin = bool(torch.isfinite(a).all())
    print("chain %s %s %s dense=%s p=%d n=%d J=%d, %d steps: pipelined twice + step-by-step, %d mismatches, finite %s, warm at %d of %d samples (t_end %.4f)"
          % (dtype, update, kw, dense, p, n, J, nsteps, bad, fin, wa, nsteps // 100, sum(h[0] for h in ha)), flush=True)
    return bad + int(not fin)


bad = 0
bad += screen(256, 256, 65536, 3000, "eks", {})
bad += screen(256, 256, 65536, 3000, "aldi", dict(time_step="constant", delta_t=0.03))
bad += screen(256, 256, 65536, 2000, "aldi", dict(time_step="mix", delta_t=0.03, spinup=1.0))
bad += screen(256, 256, 16384, 1500, "eks", {}, dense=True)
bad += screen(64, 50, 8192, 5000, "eks", {})
bad += screen(256, 256, 16384, 800, "eks", dict(time_step="constant", delta_t=0.03), "float64")
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
