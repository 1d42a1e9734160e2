#!/usr/bin/env python3
"""Golden vectors for the host forward models and for a 'pde'-type run.

    python -m oracle.make_golden_models        # build container only (needs /root/reference)

Inputs and outputs of the REAL reference: ``ces/utils.py`` (imports unmodified) for every
forward-model class, and ``sampling.run`` of ``ces/calibrate.py`` (through oracle/_refload.py)
driving ``lorenz63`` as a ``type == 'pde'`` model for a few iterations.  Only data is stored
(tests/golden/models.npz, tests/golden/pde_run.npz); no reference source enters the repo.
"""
import json
import os

import numpy as np

from . import _refload

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def make_models(ru):
    rng = np.random.default_rng(77)
    g = {}
    # --- maps -------------------------------------------------------------------------
    A = rng.standard_normal((4, 3))
    th = rng.standard_normal(3)
    g["lineal_A"], g["lineal_theta"] = A, th
    g["lineal_out"] = ru.lineal(A, b=0.25)(th)
    # (lineal_log, elliptic and banana of ces/utils.py:33-122 are outside the hot-path scope -- SURVEY.md section 2 -- and
    #  are not shipped by ces_amd.models: no fixtures for them.  The draw below stays so that the fixtures behind it
    #  keep the values of the rounds before.)
    rng.standard_normal((2, 6))
    # --- lorenz 63 --------------------------------------------------------------------
    l63 = ru.lorenz63(l_window=2, freq=25)
    t63 = np.linspace(0, 6, 151)                       # 150 samples after t0 = 3 windows of 50
    w0 = np.array([1.0, -2.0, 20.0])
    ws = l63.solve(w0, t63, args=(27.0, 2.5))
    g["l63_t"], g["l63_w0"], g["l63_ws"] = t63, w0, ws
    g["l63_rhs"] = np.asarray(l63(w0, 0.0, 27.0, 2.5))
    g["l63_stats"] = l63.statistics(ws)
    l63l = ru.lorenz63_log(l_window=2, freq=25)
    g["l63log_rhs"] = np.asarray(l63l(w0, 0.0, np.log(27.0), np.log(2.5)))
    g["l63log_ws"] = l63l.solve(w0, t63, args=(np.log(27.0), np.log(2.5)))
    # --- lorenz 96 --------------------------------------------------------------------
    l96 = ru.lorenz96(n_slow=8, n_fast=4, l_window=1, freq=10, spinup=1)
    x = rng.standard_normal(l96.n_state)
    g["l96_x"] = x
    g["l96_rhs"] = l96.model(x, 0.0, 0.9, 9.0, np.log(8.0), 7.0)
    l96.set_solver(T=3.0, dt=0.05)
    t96 = np.linspace(0, 3.0, 31)                      # 31 samples; spin-up drops 11 -> 2 windows of 10
    ws96 = l96.solve(x, t96, args=(0.9, 9.0, np.log(8.0), 7.0))
    g["l96_t"], g["l96_ws"] = t96, ws96
    g["l96_stats"] = l96.statistics(ws96)
    full = ru.lorenz96()
    xf = rng.standard_normal(full.n_state)
    g["l96full_x"] = xf
    g["l96full_rhs"] = full(0.0, xf)
    g["l96Fc_rhs"] = ru.lorenz96Fc()(0.0, xf, 8.0, np.log(9.0))
    g["l96Fb_rhs"] = ru.lorenz96Fb()(0.0, xf, 8.0, 9.0)
    g["l96hFb_rhs"] = ru.lorenz96hFb()(0.0, xf, 0.8, 8.0, 9.0)
    g["l96hcb_rhs"] = ru.lorenz96hcb()(0.0, xf, 0.8, np.log(9.0), 9.0)
    g["l96dim_rhs"] = ru.lorenz96_dim(0.0, xf)
    hom = ru.lorenz96_hom()
    hom.set_solver(T=20.0, dt=0.1)
    th = np.linspace(0, 20.0, 201)                     # 201 samples - 101 of spin-up = 1 window of 100
    wsh = hom.solve(xf, th, args=())
    g["l96hom_t"] = th                                 # (the 201 x 396 trajectory itself is not stored)
    g["l96hom_stats"] = hom.statistics(wsh)
    g["l96_grad_logjac"] = full.grad_logjacobian(np.array([1.0, 2.0, 3.0, 4.0]))
    np.random.seed(5)
    g["l96_initial"] = ru.lorenz96(n_slow=6, n_fast=3).generate_initial()
    reprs = {k: repr(v) for k, v in dict(lineal=ru.lineal(A), lorenz63=l63, lorenz63_log=l63l, lorenz96=full,
                                         lorenz96Fc=ru.lorenz96Fc(), lorenz96Fb=ru.lorenz96Fb(),
                                         lorenz96hFb=ru.lorenz96hFb(), lorenz96hcb=ru.lorenz96hcb()).items()}
    return g, reprs


def make_pde_run(ref, ru):
    """sampling.run with a 'pde' model: carried state W0 (ces/calibrate.py:317-327, 342-350, 390-398)."""
    p, J, T = 2, 12, 3
    model = ru.lorenz63(l_window=1, freq=20)
    t = np.linspace(0, 2, 41)                          # 40 samples after t0 = 2 windows of 20
    rng = np.random.default_rng(3)
    wt = np.array([2.0, 3.0, 25.0])
    ustar = np.array([[28.0], [8.0 / 3]])
    y = model.statistics(model.solve(wt, t, args=(28.0, 8.0 / 3)))
    Gamma = np.diag(0.05 * np.abs(y) + 0.1)
    mu = np.array([[26.0], [2.2]])
    sigma = np.diag([2.0 ** 2, 0.4 ** 2])
    U0 = mu + np.sqrt(sigma) @ rng.standard_normal((p, J))
    out = {}
    for update in ("aldi", "eks"):
        eks = ref.sampling(p=p, n_obs=model.n_obs, J=J)
        eks.ustar, eks.mu, eks.sigma, eks.T = ustar, mu, sigma, T
        eks.parallel, eks.mute_bar = False, True
        np.random.seed(11)
        eks.run(y, np.copy(U0), model, Gamma, np.linalg.cholesky(Gamma), wt=wt, t=t, update=update, t_tol=1e9)
        out[update + "_Ustar"] = eks.Ustar
        out[update + "_Gstar"] = eks.Gstar
        out[update + "_W0"] = eks.W0
        out[update + "_Uall"] = eks.Uall
        out[update + "_Gall"] = eks.Gall
        for k, v in eks.metrics.items():
            out[update + "_metric_" + k] = np.asarray(v)
    out.update(dict(y=y, Gamma=Gamma, mu=mu, sigma=sigma, ustar=ustar, U0=U0, wt=wt, t=t, seed=np.array(11),
                    l_window=np.array(1), freq=np.array(20), T=np.array(T)))
    return out


def main():
    import scipy
    ru = _refload.load_reference_utils()
    ref = _refload.load_reference_calibrate()
    g, reprs = make_models(ru)
    np.savez_compressed(os.path.join(OUT, "models.npz"), **g)
    np.savez_compressed(os.path.join(OUT, "pde_run.npz"), **make_pde_run(ref, ru))
    with open(os.path.join(OUT, "models_manifest.json"), "w") as f:
        json.dump(dict(generator="oracle/make_golden_models.py", numpy=np.__version__, scipy=scipy.__version__,
                       reprs=reprs, note="outputs of /root/reference/ces/utils.py and of sampling.run "
                                         "(ces/calibrate.py) with a type='pde' model"), f, indent=1)
    print("wrote models.npz (%d arrays), pde_run.npz" % len(g))


if __name__ == "__main__":
    main()
