"""dev tool: tiny ensembles through the small update kernels against the oracle (J = 4 ... 12: one ragged workgroup)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ces_amd import engine
from oracle import ces_numpy as oc
bad = 0
for (p, n, J) in ((5, 3, 8), (64, 64, 4), (17, 9, 12), (3, 2, 4), (64, 50, 68)):
    rng = np.random.default_rng(p + n + J)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    Gamma, sigma, mu = 0.01 * np.eye(n), 100.0 * np.eye(p), np.zeros((p, 1))
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    U0 = ustar + rng.standard_normal((p, J))
    G = A @ U0
    xi = rng.standard_normal((p, J))
    for update in ("aldi", "eks", "aldi_constant"):
        for dtype, tol in (("float64", 1e-6), ("float32", 2e-3)):
            if J <= p and dtype == "float32":
                # (a rank-deficient ensemble covariance: only + 1e-8 I keeps it positive definite -- below the rounding of moments
                #  taken from fp32 data; the engine reports the failed factorisation as numpy.linalg.LinAlgError there)
                continue
            st = oc.OracleState(p, n, J, mu, sigma, ustar)
            try:
                ref = oc.factored_step(st, y, U0, G, Gamma, xi, update=update)
            except Exception as ex:
                print(p, n, J, update, "oracle:", repr(ex)); continue
            eng = engine.Engine(p, n, J, dtype=dtype)
            eng.set_problem(y, Gamma, mu, sigma, ustar)
            try:
                out = eng.step(engine.step_params(update=update), U0, G, xi=xi).cpu().numpy().astype(np.float64)
                eng.result()
            except Exception as ex:
                print(p, n, J, update, dtype, "engine:", repr(ex)); bad += 1; continue
            err = np.max(np.abs(out - ref)) / max(1e-300, np.max(np.abs(ref)))
            ok = err < tol
            bad += int(not ok)
            print("p=%d n=%d J=%d %-13s %s rel err %.3e %s" % (p, n, J, update, dtype, err, "ok" if ok else "FAIL"), flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
