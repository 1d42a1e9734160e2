#!/usr/bin/env python3
"""dev tool: where the host-array update call of a SMALL ensemble spends its time (the sizes examples/scripts/darcy-flow.py
and the notebooks run: J in 17 .. 768, p = 64 / 256), stage by stage, next to the CPU oracle's literal step.

    python tools/small_j_probe.py [--iters 60]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def med(v):
    return float(np.median(np.array(v))) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    args = ap.parse_args()
    from ces_amd import engine
    from ces_amd.calibrate import sampling
    from oracle import ces_numpy as oc
    for (p, n, J) in ((64, 50, 512), (256, 50, 768), (64, 50, 8192)):
        rng = np.random.default_rng(5)
        A = rng.standard_normal((n, p)) / np.sqrt(p)
        ustar = rng.standard_normal((p, 1))
        Gamma, sigma, mu = 0.01 * np.eye(n), 100.0 * np.eye(p), np.zeros((p, 1))
        y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
        U0 = ustar + rng.standard_normal((p, J))
        for noise in ("device", "numpy"):
            eks = sampling(p=p, n_obs=n, J=J)
            eks.ustar, eks.mu, eks.sigma = ustar, mu, sigma
            eks.engine_dtype, eks.noise, eks.T = "float32", noise, 30
            eks.Uall = [U0, U0]           # (not the first step)
            eks._ensure_metrics()
            eks.metrics["t"].append(0.1)
            U = U0
            tt = []
            stages = {}
            eng = eks._get_engine()
            # stage probes (separate loop): set_problem, H2D, step call, result, D2H
            for it in range(args.iters):
                G = A @ U
                t0 = time.perf_counter(); eng.set_problem(y, Gamma, mu, sigma, ustar); t1 = time.perf_counter()
                Ud = eng.to_device(U, p, "U"); Gd = eng.to_device(G, n, "G"); t2 = time.perf_counter()
                prm = engine.step_params(update="aldi", first_step=False, t_len=1, t_last=0.1, step_index=it)
                out = eng.step(prm, Ud, Gd, xi=None, recenter=True); t3 = time.perf_counter()
                res = eng.result(); t4 = time.perf_counter()
                Uh = eng.to_host(out); t5 = time.perf_counter()
                for k, v in (("set_problem", t1 - t0), ("h2d", t2 - t1), ("step_call", t3 - t2), ("result_wait", t4 - t3), ("d2h", t5 - t4)):
                    stages.setdefault(k, []).append(v)
            for it in range(args.iters):
                G = A @ U
                t0 = time.perf_counter()
                Un = eks.eks_update_aldi(y, U, G, Gamma, it)
                tt.append(time.perf_counter() - t0)
                U = Un if it % 7 else U0            # (keep the ensemble from collapsing over many probes)
            print("p=%d n=%d J=%d noise=%s: update call median %.3f ms (min %.3f) | stages: %s" % (
                p, n, J, noise, med(tt), min(tt) * 1e3, ", ".join("%s %.3f" % (k, med(v)) for k, v in stages.items())), flush=True)
        if J <= 1024:
            st = oc.OracleState(p, n, J, mu, sigma, ustar)
            st.metrics["t"].append(0.1)
            tl = []
            for it in range(10):
                G = A @ U0
                xi = rng.standard_normal((p, J))
                t0 = time.perf_counter()
                oc.literal_step(st, y, U0, G, Gamma, xi, update="aldi")
                tl.append(time.perf_counter() - t0)
            print("   CPU oracle literal_step (ces/calibrate.py:451-490 as written): median %.3f ms" % med(tl), flush=True)


if __name__ == "__main__":
    main()
