"""Generate tests/golden/*.npz by running the REAL reference (build container only).

    python -m oracle.make_golden            # from the repo root

Every fixture is data: seeded inputs (including the captured noise block xi and
the pre-step object state) and the outputs the reference's own
``sampling.eks_update*`` / ``sampling.run`` produced for them
(ces/calibrate.py:270-529, loaded by oracle/_refload.py).  No reference source
text is stored.  The xi capture is valid because each update draws exactly one
``np.random.normal(0, 1, [p, J])`` block and nothing else consumes the global
stream when the forward map is noise free: re-seeding and
re-drawing after the call reproduces the block the update used.
"""
import io
import itertools
import json
import os
import sys
import contextlib

import numpy as np
import scipy

from . import _refload

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _spd(rng, k, scale):
    B = rng.standard_normal((k, k))
    return scale * (B @ B.T / k + np.eye(k))


def _problem(rng, p, n, J, dense_gamma, dense_sigma, nonlinear, u_scale=1.0):
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    Gamma = _spd(rng, n, 0.05) if dense_gamma else 0.05 * np.eye(n)
    sigma = _spd(rng, p, 4.0) if dense_sigma else 4.0 * np.eye(p)
    mu = 0.3 * rng.standard_normal((p, 1))
    U0 = ustar + u_scale * rng.standard_normal((p, J))
    G = A @ U0
    if nonlinear:
        G = G + 0.1 * np.sin(G) + 0.05 * (A @ U0 ** 2)
    y = (A @ ustar).ravel() + 0.2 * rng.standard_normal(n)
    return dict(A=A, ustar=ustar, Gamma=Gamma, sigma=sigma, mu=mu, U0=U0, G=G, y=y)


def _fresh(ref, p, n, J, prob, T=30):
    obj = ref.sampling(p=p, n_obs=n, J=J)
    obj.T = T
    obj.mu, obj.sigma, obj.ustar = prob["mu"], prob["sigma"], prob["ustar"]
    obj.radspec = []
    obj.metrics = {k: [] for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t")}
    return obj


def _run_step(ref, update, p, n, J, prob, kwargs, trace_len, t_prev, seed):
    obj = _fresh(ref, p, n, J, prob)
    obj.Uall = [None] * trace_len
    obj.metrics["t"] = list(t_prev)
    fn = {"eks": obj.eks_update, "aldi": obj.eks_update_aldi,
          "aldi_constant": obj.eks_update_aldi_constant}[update]
    U0 = prob["U0"].copy()
    G = prob["G"].copy()
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        Uk = fn(prob["y"], U0, G, prob["Gamma"], 0, **kwargs)
    np.random.seed(seed)
    xi = np.random.normal(0, 1, [p, J])
    assert np.array_equal(U0, prob["U0"]) and np.array_equal(G, prob["G"])
    hk = obj.metrics["t"][-1] - (0.0 if trace_len == 1 else t_prev[-1])
    return dict(Uk=Uk, xi=xi, hk=np.float64(hk), t_new=np.float64(obj.metrics["t"][-1]),
                metrics=np.array([obj.metrics[k][-1] for k in
                                  ("self-bias", "self-bias-data", "bias-data", "bias")]),
                radspec=np.array(obj.radspec, dtype=np.float64),
                update_rule=obj.update_rule)


def make_steps(ref):
    rng = np.random.default_rng(20261003)
    arrays, manifest = {}, []
    shapes = [(6, 5, 48), (3, 6, 40), (8, 4, 64)]
    ts_cases = [
        ("default", {}, [0.3]),
        ("spectral", {"time_step": "spectral"}, [0.3]),
        ("constant", {"time_step": "constant", "delta_t": 0.02}, [0.3]),
        ("constant_default_dt", {"time_step": "constant"}, [0.3]),
        ("mix_spinup", {"time_step": "mix", "spinup": 4.0, "delta_t": 0.03}, [0.2]),
        ("mix_spinup_late", {"time_step": "mix", "spinup": 4.0, "delta_t": 0.03}, [1.7]),
        ("mix_after", {"time_step": "mix", "spinup": 4.0, "delta_t": 0.03}, [4.5]),
    ]
    cid = 0
    for update in ("aldi", "eks"):
        for (tsname, kw, tprev), (dg, ds), first in itertools.product(
                ts_cases, [(0, 0), (1, 0), (0, 1), (1, 1)], (True, False)):
            if first and tsname in ("mix_spinup_late", "mix_after", "constant_default_dt"):
                continue
            p, n, J = shapes[cid % len(shapes)]
            nonlinear = (cid % 2 == 1)
            prob = _problem(rng, p, n, J, dg, ds, nonlinear, u_scale=[1.0, 0.3, 3.0][cid % 3])
            trace_len = 1 if first else 3
            t_prev = [] if first else tprev
            out = _run_step(ref, update, p, n, J, prob, kw, trace_len, t_prev, seed=1000 + cid)
            _store(arrays, manifest, cid, update, tsname, kw, p, n, J, dg, ds, nonlinear,
                   trace_len, t_prev, prob, out)
            cid += 1
    for (dg, ds), first, sw in itertools.product([(0, 0), (1, 0), (0, 1), (1, 1)],
                                                 (True, False), (None, 0.5)):
        p, n, J = shapes[cid % len(shapes)]
        nonlinear = (cid % 2 == 1)
        prob = _problem(rng, p, n, J, dg, ds, nonlinear)
        kw = {} if sw is None else {"switch": sw}
        trace_len = 1 if first else 2
        t_prev = [] if first else [0.11]
        out = _run_step(ref, "aldi_constant", p, n, J, prob, kw, trace_len, t_prev, seed=1000 + cid)
        _store(arrays, manifest, cid, "aldi_constant", "n/a", kw, p, n, J, dg, ds, nonlinear,
               trace_len, t_prev, prob, out)
        cid += 1
    return arrays, manifest


def _store(arrays, manifest, cid, update, tsname, kw, p, n, J, dg, ds, nonlinear,
           trace_len, t_prev, prob, out):
    tag = "c%03d_" % cid
    for k in ("U0", "G", "y", "Gamma", "sigma", "mu", "ustar"):
        arrays[tag + k] = prob[k]
    for k in ("Uk", "xi", "hk", "t_new", "metrics", "radspec"):
        arrays[tag + k] = out[k]
    manifest.append(dict(id=cid, update=update, time_step_case=tsname, kwargs=kw, p=p, n_obs=n, J=J,
                         dense_gamma=bool(dg), dense_sigma=bool(ds), nonlinear_G=bool(nonlinear),
                         trace_len=trace_len, t_prev=list(t_prev), update_rule=out["update_rule"]))


def make_errors(ref):
    """Failure paths of ces/calibrate.py: rank-deficient ensemble -> LinAlgError
    at :487 (after the diagnostic print at :477-480); time_step='adaptive' ->
    AttributeError at :255; unknown time_step -> UnboundLocalError at :262."""
    rng = np.random.default_rng(7)
    res, arrays = [], {}
    p, n = 6, 4
    s = 2.0 ** 20
    # Rank-1 ensembles (every parameter row identical) whose covariance is
    # EXACTLY s^2 * ones(p, p) in floating point (mean 0, sum v^2 = 4 s^2,
    # divisor 4: J-1 for aldi at J=5, J for eks at J=4) and large enough that
    # the reference's +1e-8 jitter (:424, :476) is absorbed (ulp(2^40) = 2.4e-4).
    # Cholesky's second pivot is then exactly 0 -> LinAlgError.
    cases = [("aldi", {}, np.array([-s, -s, 0.0, s, s])),
             ("eks", {}, np.array([-s, -s, s, s])),
             ("aldi_constant", {}, np.array([-s, -s, 0.0, s, s])),
             ("aldi", {"time_step": "adaptive"}, None),
             ("aldi", {"time_step": "bogus"}, None)]
    for k, (update, kw, v) in enumerate(cases):
        J = 8 if v is None else len(v)
        prob = _problem(rng, p, n, J, 0, 0, False)
        if v is not None:
            prob["U0"] = np.tile(v, (p, 1))
            prob["G"] = prob["A"] @ (prob["U0"] / s)
        try:
            _run_step(ref, update, p, n, J, prob, kw, 1, [], seed=5)
            err = None
        except Exception as exc:           # noqa: BLE001 - recording the type is the point
            err = type(exc).__name__
        res.append(dict(id=k, update=update, kwargs=kw, J=J, p=p, n_obs=n, error=err))
        for key in ("U0", "G", "y", "Gamma", "sigma", "mu", "ustar"):
            arrays["e%d_%s" % (k, key)] = prob[key]
    return arrays, res


def make_trajectories(ref, utils):
    """Config C1 (BASELINE.json configs[0]): linear-Gaussian problem of
    examples/notebooks/linear.ipynb:66-76 driven through the reference's own
    ``sampling.run`` (ces/calibrate.py:270-416) with ``utils.lineal``
    (ces/utils.py:5-31), J=100, p=2, n_obs=10, T=30."""
    n_obs, p, J, T = 10, 2, 100, 30
    u_star = np.array([[-1.0], [2.0]])
    noise = 0.1
    Gamma = noise * np.eye(n_obs)
    np.random.seed(1)
    A = np.concatenate([np.ones([n_obs, 1]), 2 * np.random.normal(0, 1, [n_obs, 1])], axis=1)
    y_obs = (A @ u_star + np.sqrt(noise) * np.random.normal(0, 1, [n_obs, 1])).flatten()
    arrays = dict(A=A, y=y_obs, Gamma=Gamma, ustar=u_star,
                  mu=np.zeros((p, 1)), sigma=100.0 * np.eye(p))
    info = []
    for update in ("aldi", "eks", "aldi_constant"):
        model = utils.lineal(A)
        obj = ref.sampling(p=p, n_obs=n_obs, J=J)
        obj.T = T
        obj.ustar, obj.mu, obj.sigma = u_star, arrays["mu"], arrays["sigma"]
        obj.mute_bar = True
        seed_u0, seed_run = 11, 12
        np.random.seed(seed_u0)
        U0 = np.random.normal(0, 1, [p, J])
        np.random.seed(seed_run)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            obj.run(y_obs, U0, model, Gamma, np.linalg.cholesky(Gamma),
                    update=update, t_tol=1e9)
        nsteps = len(obj.metrics["t"])
        np.random.seed(seed_run)
        xis = np.stack([np.random.normal(0, 1, [p, J]) for _ in range(nsteps)])
        tag = update + "_"
        arrays[tag + "U0"] = U0
        arrays[tag + "xis"] = xis
        arrays[tag + "Uall"] = np.asarray(obj.Uall)
        arrays[tag + "Gall"] = np.asarray(obj.Gall)
        for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t"):
            arrays[tag + "m_" + k] = np.asarray(obj.metrics[k], dtype=np.float64)
        info.append(dict(update=update, steps=nsteps, J=J, p=p, n_obs=n_obs, T=T,
                         seed_u0=seed_u0, seed_run=seed_run, update_rule=obj.update_rule,
                         repr_fresh=repr(ref.sampling(p=p, n_obs=n_obs, J=40))))
    return arrays, info


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ref = _refload.load_reference_calibrate()
    utils = _refload.load_reference_utils()
    versions = dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0])

    arrays, manifest = make_steps(ref)
    np.savez_compressed(os.path.join(GOLDEN, "steps.npz"), **arrays)
    earr, errors = make_errors(ref)
    np.savez_compressed(os.path.join(GOLDEN, "errors.npz"), **earr)
    tarr, tinfo = make_trajectories(ref, utils)
    np.savez_compressed(os.path.join(GOLDEN, "trajectories.npz"), **tarr)
    with open(os.path.join(GOLDEN, "manifest.json"), "w") as fh:
        json.dump(dict(generator="oracle/make_golden.py", versions=versions, steps=manifest,
                       errors=errors, trajectories=tinfo), fh, indent=1)
    print("steps: %d cases, errors: %s, trajectories: %s" %
          (len(manifest), [e["error"] for e in errors], [(t["update"], t["steps"]) for t in tinfo]))


if __name__ == "__main__":
    main()
