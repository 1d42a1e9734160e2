"""numpy CPU restatement of the reference's EKS / ALDI ensemble update.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity: PINNED against the real
reference through tests/golden/ (oracle/make_golden.py).

Two forms are provided, both taking the noise block ``xi`` as an input because
the reference draws it from numpy's global legacy RNG
(ces/calibrate.py:447, 488, 527):

``literal_*``   follows ces/calibrate.py line by line and forms the J x J
                matrix D (ces/calibrate.py:429, 461, 503) -- O(n J^2).
``factored_*``  algebraically identical J x J-free form:
                everything is expressed through (p+n) x (p+n) second moments,
                O((p+n)^2 J).  This is what the HIP engine computes and what
                bench.py times as the CPU baseline at full J.

Symbols: U (p, J) ensemble, G (n, J) forward-map outputs, y (n,), Gamma (n, n),
mu (p, 1), sigma (p, p), ustar (p, 1), xi (p, J).
"""
import numpy as np

UPDATES = ("eks", "aldi", "aldi_constant")
METRIC_KEYS = ("self-bias", "self-bias-data", "bias-data", "bias")


class OracleState:
    """The slice of ``sampling`` object state the update reads and writes.

    Mirrors ces/calibrate.py:14-22 (ctor defaults), :329-339 (metrics dict and
    ``radspec``) and the ``len(self.Uall)`` first-step test at :262 / :520
    (``trace_len`` here).
    """

    def __init__(self, p, n_obs, J, mu, sigma, ustar, T=30):
        self.p, self.n_obs, self.J, self.T = int(p), int(n_obs), int(J), T
        self.mu = np.asarray(mu, dtype=np.float64).reshape(self.p, 1)
        self.sigma = np.asarray(sigma, dtype=np.float64)
        self.ustar = np.asarray(ustar, dtype=np.float64).reshape(self.p, 1)
        self.metrics = {k: [] for k in METRIC_KEYS}
        self.metrics["t"] = []
        self.radspec = []
        self.trace_len = 1
        self.update_rule = None

    def _advance_time(self, hk):
        # ces/calibrate.py:262-265 and :520-523
        if self.trace_len == 1:
            self.metrics["t"].append(hk)
        else:
            self.metrics["t"].append(hk + self.metrics["t"][-1])


# --------------------------------------------------------------------------
# literal form
# --------------------------------------------------------------------------

def _literal_D(E, R, A, J):
    # ces/calibrate.py:429 -- D = (1/J) E^T A^{-1} R
    return (1.0 / J) * (E.T @ np.linalg.solve(A, R))


def literal_timestep(st, D, **kwargs):
    """ces/calibrate.py:243-267."""
    rule = kwargs.get("time_step", None)
    if rule is None:
        hk = 1.0 / (np.linalg.norm(D) + 1e-8)
    elif rule == "spectral":
        st.radspec.append(np.linalg.eigvals(D).real.max())
        hk = 1.0 / st.radspec[-1]
    elif rule == "constant":
        hk = kwargs.get("delta_t", 1.0 / (st.T / 2))
    elif rule == "adaptive":
        # :255 calls self.LM_procedure, which is defined nowhere in the reference
        raise AttributeError("'sampling' object has no attribute 'LM_procedure'")
    elif rule == "mix":
        t = st.metrics["t"]
        if len(t) == 0 or t[-1] < kwargs.get("spinup", 4.0):
            hk = 1.0 / (np.linalg.norm(D) + 1e-8)
        else:
            hk = kwargs.get("delta_t", 1.0 / (st.T / 2))
    else:
        raise UnboundLocalError("local variable 'hk' referenced before assignment")
    st._advance_time(hk)
    return hk


def _literal_metrics(st, U0, E, R, Gamma):
    # ces/calibrate.py:432-435 (= :464-467, :506-509)
    m = st.metrics
    Ubar = U0.mean(axis=1)[:, None]
    m["self-bias"].append(((U0 - Ubar) ** 2).sum(axis=0).mean())
    m["bias"].append(((U0 - st.ustar) ** 2).sum(axis=0).mean())
    m["self-bias-data"].append((np.diag(E.T @ np.linalg.solve(Gamma, E)) ** 2).mean())
    m["bias-data"].append((np.diag(R.T @ np.linalg.solve(Gamma, R)) ** 2).mean())


def literal_step(st, y_obs, U0, Geval, Gamma, xi, update="aldi", **kwargs):
    """One ensemble update, literal form.  Returns a new (p, J) array.

    update='eks'           ces/calibrate.py:418-449
    update='aldi'          ces/calibrate.py:451-490
    update='aldi_constant' ces/calibrate.py:492-529
    """
    U0 = np.asarray(U0, dtype=np.float64)
    Geval = np.asarray(Geval, dtype=np.float64)
    y_obs = np.asarray(y_obs, dtype=np.float64)
    p, J = st.p, st.J
    rule = kwargs.get("time_step", None)

    E = Geval - Geval.mean(axis=1)[:, None]
    R = Geval - y_obs[:, None]
    D = _literal_D(E, R, Gamma, J)
    _literal_metrics(st, U0, E, R, Gamma)

    if update == "eks":
        st.update_rule = "eks_update"
        Ubar = U0.mean(axis=1)[:, None]
        C = np.cov(U0, bias=True) + 1e-8 * np.identity(p)
        hk = literal_timestep(st, D, **kwargs)
        if rule in ("adaptive", "constant"):
            Cgg = np.cov(Geval, bias=True)
            D = _literal_D(E, R, hk * Cgg + Gamma, J)
        lhs = np.eye(p) + hk * np.linalg.solve(st.sigma.T, C.T).T
        rhs = U0 - hk * ((U0 - Ubar) @ D) + hk * (C @ np.linalg.solve(st.sigma, st.mu))
        return np.linalg.solve(lhs, rhs) + np.sqrt(2 * hk) * (np.linalg.cholesky(C) @ xi)

    if update == "aldi":
        st.update_rule = "eks_update_linear"
        hk = literal_timestep(st, D, **kwargs)
        if rule in ("adaptive", "constant") or (rule == "mix" and st.metrics["t"][-1] > 1):
            Cgg = np.cov(Geval, bias=True)
            D = _literal_D(E, R, hk * Cgg + Gamma, J)
        Ubar = U0.mean(axis=1)[:, None]
        C = np.cov(U0) + 1e-8 * np.identity(p)
        alpha = (p + 1.0) / J
        return (U0 - hk * ((U0 - Ubar) @ D)
                - hk * (C @ np.linalg.solve(st.sigma, U0 - st.mu))
                + hk * alpha * (U0 - Ubar)
                + np.sqrt(2 * hk) * (np.linalg.cholesky(C) @ xi))

    if update == "aldi_constant":
        st.update_rule = "eks_update_aldi"
        Ubar = U0.mean(axis=1)[:, None]
        C = np.cov(U0) + 1e-8 * np.identity(p)
        alpha = (p + 1.0) / J
        drift = (-((U0 - Ubar) @ D)
                 - C @ np.linalg.solve(st.sigma, U0 - st.mu)
                 + kwargs.get("switch", 1.0) * alpha * (U0 - Ubar))
        hk = 0.1 / np.max(np.abs(drift))
        st._advance_time(hk)
        return U0 + hk * drift + np.sqrt(2 * hk) * (np.linalg.cholesky(C) @ xi)

    raise ValueError("unknown update rule %r" % (update,))


# --------------------------------------------------------------------------
# factored (J x J-free) form
# --------------------------------------------------------------------------

def moments(U0, Geval, acc=np.float64):
    """Ensemble means and centred second moments.

    Returns ubar (p,1), gbar (n,1), S_uu (p,p), S_ug (p,n), S_ee (n,n), all in
    ``acc`` precision; the products run in the dtype of the inputs.
    """
    ubar = U0.mean(axis=1, dtype=acc)[:, None]
    gbar = Geval.mean(axis=1, dtype=acc)[:, None]
    Au = U0 - ubar.astype(U0.dtype)
    E = Geval - gbar.astype(Geval.dtype)
    return (ubar, gbar, (Au @ Au.T).astype(acc), (Au @ E.T).astype(acc),
            (E @ E.T).astype(acc), Au, E)


def factored_timestep(st, S_ee, S_rr, Gamma, **kwargs):
    """ces/calibrate.py:243-267 with ||D||_F and eig(D) taken from n x n moments."""
    J = st.J
    rule = kwargs.get("time_step", None)

    def frob():
        X = np.linalg.solve(Gamma, S_rr)          # Gamma^{-1} S_rr
        X = np.linalg.solve(Gamma, X.T).T         # ... Gamma^{-T}
        return np.sqrt(max(float((X * S_ee).sum()), 0.0)) / J

    if rule is None:
        hk = 1.0 / (frob() + 1e-8)
    elif rule == "spectral":
        # eig(D) \ {0} = eig(Gamma^{-1} S_ee / J) \ {0} (because R E^T = S_ee);
        # D (J x J, rank <= J-1) always has a zero eigenvalue and the n x n
        # matrix is similar to an SPSD one, so the maximum is max(lam_max, 0)
        lam = np.linalg.eigvals(np.linalg.solve(Gamma, S_ee) / J).real.max()
        st.radspec.append(max(float(lam), 0.0))
        hk = 1.0 / st.radspec[-1]
    elif rule == "constant":
        hk = kwargs.get("delta_t", 1.0 / (st.T / 2))
    elif rule == "adaptive":
        raise AttributeError("'sampling' object has no attribute 'LM_procedure'")
    elif rule == "mix":
        t = st.metrics["t"]
        if len(t) == 0 or t[-1] < kwargs.get("spinup", 4.0):
            hk = 1.0 / (frob() + 1e-8)
        else:
            hk = kwargs.get("delta_t", 1.0 / (st.T / 2))
    else:
        raise UnboundLocalError("local variable 'hk' referenced before assignment")
    st._advance_time(hk)
    return hk


def factored_step(st, y_obs, U0, Geval, Gamma, xi, update="aldi", dtype=np.float64, **kwargs):
    """One ensemble update without any J x J object.  ``dtype`` is the
    arithmetic type of the O(J) passes (moments, update GEMM); the small dense
    algebra always runs in float64.  Returns a new (p, J) array of ``dtype``.
    """
    dt = np.dtype(dtype)
    U0 = np.ascontiguousarray(U0, dtype=dt)
    Geval = np.ascontiguousarray(Geval, dtype=dt)
    xi = np.asarray(xi, dtype=dt)
    y = np.asarray(y_obs, dtype=np.float64).reshape(-1, 1)
    Gamma = np.asarray(Gamma, dtype=np.float64)
    p, J = st.p, st.J
    rule = kwargs.get("time_step", None)

    ubar, gbar, S_uu, S_ug, S_ee, Au, E = moments(U0, Geval)
    m = gbar - y
    S_rr = S_ee + J * (m @ m.T)

    # metrics (ces/calibrate.py:432-435); the two data metrics are 4th order
    # in the particles and need one pass over G
    met = st.metrics
    met["self-bias"].append(float(np.trace(S_uu)) / J)
    met["bias"].append(float(np.trace(S_uu)) / J + float(((ubar - st.ustar) ** 2).sum()))
    Ginv = np.linalg.inv(Gamma)
    E64 = E.astype(np.float64)
    R64 = E64 + m
    met["self-bias-data"].append(float((((Ginv @ E64) * E64).sum(axis=0) ** 2).mean()))
    met["bias-data"].append(float((((Ginv @ R64) * R64).sum(axis=0) ** 2).mean()))

    def gain(A):
        # (S_ug / J) A^{-1}
        return np.linalg.solve(A.T, (S_ug / J).T).T

    if update == "aldi_constant":
        st.update_rule = "eks_update_aldi"
        C = S_uu / (J - 1) + 1e-8 * np.identity(p)
        alpha = (p + 1.0) / J
        K = gain(Gamma)
        M = np.linalg.solve(st.sigma.T, C.T).T           # C sigma^{-1}
        sw = kwargs.get("switch", 1.0) * alpha
        Wd = np.hstack([sw * np.eye(p) - M, -K]).astype(dt)
        bd = (K @ y + M @ st.mu - sw * ubar).astype(dt)
        drift = Wd @ np.vstack([U0, Geval]) + bd
        hk = 0.1 / float(np.max(np.abs(drift)))
        st._advance_time(hk)
        L = np.linalg.cholesky(C)
        return U0 + dt.type(hk) * drift + (np.sqrt(2 * hk) * L).astype(dt) @ xi

    hk = factored_timestep(st, S_ee, S_rr, Gamma, **kwargs)
    recompute = rule in ("adaptive", "constant")
    if update == "aldi":
        recompute = recompute or (rule == "mix" and st.metrics["t"][-1] > 1)
    K = gain(hk * (S_ee / J) + Gamma) if recompute else gain(Gamma)

    if update == "aldi":
        st.update_rule = "eks_update_linear"
        C = S_uu / (J - 1) + 1e-8 * np.identity(p)
        alpha = (p + 1.0) / J
        M = np.linalg.solve(st.sigma.T, C.T).T
        L = np.linalg.cholesky(C)
        W = np.hstack([(1 + hk * alpha) * np.eye(p) - hk * M, -hk * K, np.sqrt(2 * hk) * L])
        b = hk * (K @ y + M @ st.mu - alpha * ubar)
    elif update == "eks":
        st.update_rule = "eks_update"
        C = S_uu / J + 1e-8 * np.identity(p)
        M = np.linalg.solve(st.sigma.T, C.T).T
        L = np.linalg.cholesky(C)
        P = np.linalg.inv(np.eye(p) + hk * M)
        W = np.hstack([P, -hk * (P @ K), np.sqrt(2 * hk) * L])
        b = P @ (hk * (K @ y + M @ st.mu))
    else:
        raise ValueError("unknown update rule %r" % (update,))

    X = np.vstack([U0, Geval, xi])
    return W.astype(dt) @ X + b.astype(dt)


# --------------------------------------------------------------------------
# forward map + driver loop
# --------------------------------------------------------------------------

def lineal_forward(A, U, b=0.0):
    """ces/utils.py:25-31 applied to every particle (ces/calibrate.py:123-130)."""
    return np.asarray(A) @ np.asarray(U) + b


def run_chain(st, y_obs, U0, forward, Gamma, xis, update="aldi", step=literal_step, **kwargs):
    """Driver loop of ces/calibrate.py:341-408 for ``model.type == 'map'`` with
    ``trace=True``: returns (Uall, Gall) stacked like the reference's traces.
    ``xis`` is the list of injected noise blocks, one per iteration.
    """
    Uall, Gall = [], []
    t_tol = kwargs.get("t_tol", 2.0)
    U = np.asarray(U0)
    for i in range(st.T):
        G = forward(U)
        Uall.append(U)
        Gall.append(G)
        st.trace_len = len(Uall)
        U = step(st, y_obs, U, G[: st.n_obs], Gamma, xis[i], update=update, **kwargs)
        if st.metrics["t"][-1] > t_tol:
            break
    Uall.append(U)
    Gall.append(forward(U))
    return np.asarray(Uall), np.asarray(Gall)
