"""GPU parity: the HIP engine (through the C ABI) against the golden fixtures
produced by the real reference and against the CPU oracle.

Tolerances are BASELINE.json's: 1e-6 relative in fp64, 1e-3 relative in fp32
(measured errors are far below both; the tighter internal bounds used for fp64
are stated per test).
"""
import os

import numpy as np
import pytest

from conftest import step_case, rel_err

pytestmark = pytest.mark.gpu

TOL64 = 1e-6        # north star
TOL64_TIGHT = 1e-9  # what fp64 actually delivers on the small golden cases
TOL32 = 1e-3


@pytest.fixture(scope="module")
def eng_mod():
    import torch
    from ces_amd import build, engine
    build.build_lib()
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return engine


def _run_case(engine, case, c, dtype, xi="golden"):
    eng = engine.Engine(case["p"], case["n_obs"], case["J"], dtype=dtype)
    eng.set_problem(c["y"], c["Gamma"], c["mu"], c["sigma"], c["ustar"])
    kw = case["kwargs"]
    t_prev = case["t_prev"]
    prm = engine.step_params(update=case["update"], time_step=kw.get("time_step"),
                             first_step=case["trace_len"] == 1, t_len=len(t_prev),
                             t_last=t_prev[-1] if t_prev else 0.0, delta_t=kw.get("delta_t"),
                             spinup=kw.get("spinup", 4.0), switch=kw.get("switch", 1.0), T=30)
    out = eng.step(prm, c["U0"], c["G"], xi=c["xi"] if xi == "golden" else xi)
    res = eng.result()
    return out.cpu().numpy().astype(np.float64), res, eng


@pytest.mark.parametrize("dtype,tol", [("float64", TOL64_TIGHT), ("float32", TOL32)])
def test_golden_steps(eng_mod, manifest, golden_steps, dtype, tol):
    """Every golden step case of the reference: all three update rules, all time
    step rules, dense/diagonal Gamma and Sigma, linear/non-linear G, first/later step."""
    worst = {}
    for case in manifest["steps"]:
        c = step_case(golden_steps, case)
        Uk, res, _ = _run_case(eng_mod, case, c, dtype)
        err = rel_err(Uk, c["Uk"])
        key = (case["update"], case["time_step_case"])
        worst[key] = max(worst.get(key, 0.0), err)
        assert err < tol, (case, err)
        mtol = TOL32 if dtype == "float32" else 1e-8          # the north star's bar, scalars included
        assert res.t_new == pytest.approx(float(c["t_new"]), rel=mtol), case
        assert res.hk == pytest.approx(float(c["hk"]), rel=mtol, abs=1e-300), case
        got = np.array([res.self_bias, res.self_bias_data, res.bias_data, res.bias])
        assert np.allclose(got, c["metrics"], rtol=mtol), (case, got, c["metrics"])
        if case["kwargs"].get("time_step") == "spectral":
            assert res.radspec == pytest.approx(float(c["radspec"][-1]), rel=mtol), case
    print("worst relative error per (update, time step):", worst)


def test_update_returns_new_array_and_keeps_input(eng_mod, manifest, golden_steps):
    import torch
    case = manifest["steps"][0]
    c = step_case(golden_steps, case)
    eng = eng_mod.Engine(case["p"], case["n_obs"], case["J"], dtype="float64")
    eng.set_problem(c["y"], c["Gamma"], c["mu"], c["sigma"], c["ustar"])
    U = eng.to_device(c["U0"])
    before = U.clone()
    out = eng.step(eng_mod.step_params(update="aldi"), U, eng.to_device(c["G"]), xi=eng.to_device(c["xi"]))
    eng.result()
    assert out.data_ptr() != U.data_ptr() and torch.equal(U, before)     # ces/calibrate.py:357
    with pytest.raises(ValueError):
        eng.step(eng_mod.step_params(update="aldi"), U, eng.to_device(c["G"]), out=U)


def test_error_paths(eng_mod, manifest, golden_errors):
    """Rank-deficient ensemble -> LinAlgError (ces/calibrate.py:446/:487/:526);
    'adaptive' -> AttributeError (:255); unknown rule -> UnboundLocalError (:262)."""
    for e in manifest["errors"]:
        c = {k.split("_", 1)[1]: v for k, v in golden_errors.items() if k.startswith("e%d_" % e["id"])}
        expected = {"LinAlgError": np.linalg.LinAlgError, "AttributeError": AttributeError,
                    "UnboundLocalError": UnboundLocalError}[e["error"]]
        with pytest.raises(expected):
            eng = eng_mod.Engine(e["p"], e["n_obs"], e["J"], dtype="float64")
            eng.set_problem(c["y"], c["Gamma"], c["mu"], c["sigma"], c["ustar"])
            prm = eng_mod.step_params(update=e["update"], time_step=e["kwargs"].get("time_step"))
            eng.step(prm, c["U0"], c["G"], xi=np.zeros((e["p"], e["J"])))
            eng.result()


@pytest.mark.parametrize("update", ["aldi", "eks", "aldi_constant"])
def test_trajectory_c1_drop_in(eng_mod, manifest, golden_traj, update):
    """BASELINE.json configs[0] through the drop-in class: same seeds as the
    reference run -> same 30-step trajectory (noise='numpy' consumes the global
    numpy stream exactly like ces/calibrate.py:447/:488/:527)."""
    from ces_amd.calibrate import sampling
    from ces_amd.utils import lineal
    info = next(t for t in manifest["trajectories"] if t["update"] == update)
    g = golden_traj
    eks = sampling(p=info["p"], n_obs=info["n_obs"], J=info["J"])
    eks.T = info["T"]
    eks.ustar, eks.mu, eks.sigma = g["ustar"], g["mu"], g["sigma"]
    np.random.seed(info["seed_u0"])
    U0 = np.random.normal(0, 1, [info["p"], info["J"]])
    assert np.array_equal(U0, g[update + "_U0"])
    np.random.seed(info["seed_run"])
    eks.run_eks(g["y"], U0, lineal(g["A"]), g["Gamma"], np.linalg.cholesky(g["Gamma"]), update=update, t_tol=1e9)
    assert eks.Uall.shape == g[update + "_Uall"].shape
    assert rel_err(eks.Uall, g[update + "_Uall"]) < TOL64
    assert rel_err(eks.Gall, g[update + "_Gall"]) < TOL64
    for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t"):
        assert np.allclose(eks.metrics[k], g[update + "_m_" + k], rtol=1e-6), k
    assert eks.update_rule == info["update_rule"]
    assert np.array_equal(eks.Ustar, eks.Uall[-1]) and eks.Gstar.shape == (info["n_obs"], info["J"])


@pytest.mark.parametrize("shape,update,dtype,tol", [((2, 10, 100, 30), "aldi", "float64", 1e-9), ((2, 10, 100, 30), "eks", "float64", 1e-9),
                                                    ((2, 10, 100, 12), "aldi_constant", "float64", 1e-9),
                                                    ((256, 256, 4096, 5), "aldi", "float64", 1e-8),
                                                    ((256, 256, 4096, 5), "aldi", "float32", 2e-4)])
def test_device_resident_run_equals_host_array_run(eng_mod, shape, update, dtype, tol):
    """``sampling.run`` with the ensemble resident on the device (model.forward_device + injected / device noise)
    against the SAME class driven the reference's way (host forward map through G_ens, float64 numpy arrays in and
    out of every update, ces/calibrate.py:341-369), same injected noise blocks: same traces, metrics, final state.
    C1 (BASELINE.json configs[0]) and a p = n_obs = 256 problem."""
    from ces_amd.calibrate import sampling
    from ces_amd.utils import lineal
    p, n, J, T = shape
    rng = np.random.default_rng(17)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    Gamma = 0.1 * np.eye(n)
    U0 = rng.standard_normal((p, J))
    xis = rng.standard_normal((T, p, J))
    runs = []
    for device_loop in (False, True):
        eks = sampling(p=p, n_obs=n, J=J)
        eks.T, eks.ustar, eks.mu, eks.sigma = T, ustar, np.zeros((p, 1)), 100.0 * np.eye(p)
        eks.engine_dtype, eks.device_loop = dtype, device_loop
        model = lineal(A)
        if not device_loop:
            # the reference's G_ens calls the model once per particle (65 536 Python calls at J = 65 536): evaluate
            # the same map for the whole ensemble at once on the host for the larger case
            eks.G_ens = lambda theta, m: A @ theta
        eks.run(y, np.copy(U0), model, Gamma, np.linalg.cholesky(Gamma), update=update, t_tol=1e9, xis=xis)
        runs.append(eks)
    a, b = runs
    assert b.Uall.shape == a.Uall.shape == (T + 1, p, J) and b.Gall.shape == a.Gall.shape
    assert rel_err(b.Uall, a.Uall) < tol and rel_err(b.Gall, a.Gall) < tol
    assert rel_err(b.Ustar, a.Ustar) < tol and rel_err(b.Gstar, a.Gstar) < tol
    for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t"):
        assert len(b.metrics[k]) == T and np.allclose(b.metrics[k], a.metrics[k], rtol=max(tol, 1e-9) * 10), k
    assert b.update_rule == a.update_rule and b.online_path == a.online_path


def test_device_resident_run_trace_stride_and_resume(eng_mod):
    """Build-only ``trace_stride``: every k-th iterate (plus the first and the final ensemble) is copied to the
    host; metrics are complete either way; a second ``run`` resumes (ces/calibrate.py:307-310) with fresh noise."""
    from ces_amd.calibrate import sampling
    from ces_amd.utils import lineal
    rng = np.random.default_rng(3)
    p, n, J, T = 8, 6, 512, 7
    A = rng.standard_normal((n, p))
    eks = sampling(p=p, n_obs=n, J=J)
    eks.T, eks.ustar, eks.mu, eks.sigma = T, np.ones((p, 1)), np.zeros((p, 1)), 10.0 * np.eye(p)
    eks.noise, eks.trace_stride = "device", 3
    y = A @ np.ones(p)
    eks.run(y, rng.standard_normal((p, J)), lineal(A), 0.1 * np.eye(n), None, t_tol=1e9)
    assert eks.Uall.shape == (3 + 1, p, J)                    # iterates 0, 3, 6 and the final ensemble
    assert len(eks.metrics["t"]) == T and np.all(np.diff(eks.metrics["t"]) > 0)
    assert np.array_equal(eks.Ustar, eks.Uall[-1]) and eks.Ustar_device.shape == (p, J)
    first = eks.Ustar.copy()
    eks.run(y, eks.Ustar, lineal(A), 0.1 * np.eye(n), None, t_tol=1e9)
    assert len(eks.metrics["t"]) == 2 * T and eks.metrics["t"][T] > eks.metrics["t"][T - 1]
    assert eks.Uall.shape[0] == 4 + 4 and not np.allclose(eks.Ustar, first)
    # trace=False: nothing but the final ensemble comes back
    eks2 = sampling(p=p, n_obs=n, J=J)
    eks2.T, eks2.ustar, eks2.mu, eks2.sigma, eks2.noise = T, np.ones((p, 1)), np.zeros((p, 1)), 10.0 * np.eye(p), "device"
    eks2.run(y, rng.standard_normal((p, J)), lineal(A), 0.1 * np.eye(n), None, trace=False, t_tol=1e9)
    assert not hasattr(eks2, "Uall") and eks2.Ustar.shape == (p, J) and len(eks2.metrics["t"]) == T


def _synthetic(p, n, J, seed, dense=False):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    if dense:
        B = rng.standard_normal((n, n)); Gamma = 0.01 * (B @ B.T / n + np.eye(n))
        B = rng.standard_normal((p, p)); sigma = 10.0 * (B @ B.T / p + np.eye(p))
    else:
        Gamma, sigma = 0.01 * np.eye(n), 100.0 * np.eye(p)
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    U0 = ustar + 0.5 * rng.standard_normal((p, J))
    G = A @ U0 + 0.05 * np.sin(A @ U0)
    xi = rng.standard_normal((p, J))
    return dict(A=A, ustar=ustar, Gamma=Gamma, sigma=sigma, mu=np.zeros((p, 1)), y=y, U0=U0, G=G, xi=xi)


@pytest.mark.parametrize("p,n,J,dense", [(64, 50, 8192, False), (256, 256, 4096, False),
                                         (100, 37, 1001, True), (256, 256, 2048, True), (33, 65, 515, False)])
@pytest.mark.parametrize("dtype,tol", [("float64", TOL64), ("float32", TOL32)])
@pytest.mark.parametrize("update", ["aldi", "eks", "aldi_constant"])
def test_medium_sizes_against_oracle(eng_mod, p, n, J, dense, dtype, tol, update):
    """Sizes the CPU oracle finishes in seconds, including ragged J (not a
    multiple of any tile), p/n not multiples of 16/32, config C4's p=64,n=50."""
    from oracle import ces_numpy as oc
    d = _synthetic(p, n, J, seed=p + n + J, dense=dense)
    st = oc.OracleState(p, n, J, d["mu"], d["sigma"], d["ustar"])
    ref = oc.factored_step(st, d["y"], d["U0"], d["G"], d["Gamma"], d["xi"], update=update)
    eng = eng_mod.Engine(p, n, J, dtype=dtype)
    eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    out = eng.step(eng_mod.step_params(update=update), d["U0"], d["G"], xi=d["xi"])
    res = eng.result()
    assert rel_err(out.cpu().numpy(), ref) < tol
    mt = 1e-7 if dtype == "float64" else TOL32
    assert res.hk == pytest.approx(st.metrics["t"][-1], rel=mt)
    got = np.array([res.self_bias, res.self_bias_data, res.bias_data, res.bias])
    want = np.array([st.metrics[k][-1] for k in ("self-bias", "self-bias-data", "bias-data", "bias")])
    assert np.allclose(got, want, rtol=mt), (got, want)


@pytest.mark.parametrize("cond,offset", [(1e2, 1.0), (1e4, 1e2)])
@pytest.mark.parametrize("update", ["aldi", "eks"])
def test_dense_gamma_conditioning(eng_mod, cond, offset, update):
    """A dense Gamma is whitened in the ENGINE dtype (G~ = L_Gamma^{-1} G by the update kernel): on an fp32 engine the rounding of
    that product, eps32 |L^{-1}| |G|, is amplified by sqrt(cond(Gamma)) relative to the whitened signal L^{-1} (G - gbar), and a
    mean far from the spread (|gbar| / spread = offset) adds to it.  The supported range of include/cesx.h -- cond(Gamma) <= 1e4 with
    |gbar| / spread <= 1e2 -- against the pinned oracle at the fp32 bar; the reference's dense Gammas are sample covariances of a few
    observables (examples/notebooks/lorenz63.ipynb)."""
    from oracle import ces_numpy as oc
    p, n, J = 64, 48, 4096
    rng = np.random.default_rng(int(cond) + 3)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(0, -np.log10(cond), n)
    Gamma = 0.01 * (Q * ev) @ Q.T
    Gamma = 0.5 * (Gamma + Gamma.T)
    sigma, mu = 100.0 * np.eye(p), np.zeros((p, 1))
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    U0 = ustar + 0.5 * rng.standard_normal((p, J))
    G = A @ U0
    spread = G.std(axis=1).mean()
    shift = offset * spread * rng.standard_normal(n)          # a forward map with an offset: |gbar| >> the ensemble's spread
    G = G + shift[:, None]
    y = y + shift
    xi = rng.standard_normal((p, J))
    st = oc.OracleState(p, n, J, mu, sigma, ustar)
    ref = oc.factored_step(st, y, U0, G, Gamma, xi, update=update)
    eng = eng_mod.Engine(p, n, J, dtype="float32")
    eng.set_problem(y, Gamma, mu, sigma, ustar)
    out = eng.step(eng_mod.step_params(update=update), U0, G, xi=xi)
    res = eng.result()
    assert rel_err(out.cpu().numpy(), ref) < TOL32
    assert res.hk == pytest.approx(st.metrics["t"][-1], rel=TOL32)
    got = np.array([res.self_bias, res.self_bias_data, res.bias_data, res.bias])
    want = np.array([st.metrics[k][-1] for k in ("self-bias", "self-bias-data", "bias-data", "bias")])
    assert np.allclose(got, want, rtol=TOL32), (got, want)


# time-step rules at the benchmark shape (ces/calibrate.py:247-260, :439-441, :470-473): the Krylov space of
# the spectral rule is NOT exhausted at n_obs = 256 (the golden cases have n_obs <= 6), the gain recompute
# inverts a 256 x 256 (hk C_gg + Gamma), and dense Gamma takes the general path of both
_TS_CASES = {
    "spectral": (dict(time_step="spectral"), []),
    "constant_dt": (dict(time_step="constant", delta_t=0.02), [0.4]),
    "constant_default": (dict(time_step="constant"), []),                      # delta_t = 1 / (T / 2)
    "mix_spinup_done": (dict(time_step="mix", delta_t=0.05, spinup=0.5), [0.3, 0.9]),    # hk = delta_t, t_new <= 1
    "mix_late_recompute": (dict(time_step="mix", delta_t=0.05, spinup=2.0), [1.2, 2.5]),  # t_new > 1: gain recomputed (aldi)
    "mix_before_spinup": (dict(time_step="mix", delta_t=0.05, spinup=4.0), [0.2]),        # Frobenius rule
}


@pytest.mark.parametrize("ts", sorted(_TS_CASES))
@pytest.mark.parametrize("p,n,J,dense", [(256, 256, 4096, False), (256, 256, 4096, True),
                                         (64, 50, 8192, False), (64, 50, 8192, True)])
@pytest.mark.parametrize("dtype,tol", [("float64", TOL64), ("float32", TOL32)])
@pytest.mark.parametrize("update", ["aldi", "eks"])
def test_time_step_rules_at_benchmark_shape(eng_mod, ts, p, n, J, dense, dtype, tol, update):
    from oracle import ces_numpy as oc
    kw, t_prev = _TS_CASES[ts]
    d = _synthetic(p, n, J, seed=p + n + J + 7, dense=dense)
    st = oc.OracleState(p, n, J, d["mu"], d["sigma"], d["ustar"], T=30)
    st.metrics["t"] = list(t_prev)
    st.trace_len = 1 if not t_prev else 2
    ref = oc.factored_step(st, d["y"], d["U0"], d["G"], d["Gamma"], d["xi"], update=update, **kw)
    eng = eng_mod.Engine(p, n, J, dtype=dtype)
    eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    prm = eng_mod.step_params(update=update, time_step=kw["time_step"], first_step=not t_prev, t_len=len(t_prev),
                              t_last=t_prev[-1] if t_prev else 0.0, delta_t=kw.get("delta_t"),
                              spinup=kw.get("spinup", 4.0), T=30)
    out = eng.step(prm, d["U0"], d["G"], xi=d["xi"])
    res = eng.result()
    assert rel_err(out.cpu().numpy(), ref) < tol
    assert res.t_new == pytest.approx(st.metrics["t"][-1], rel=tol)
    hk_ref = st.metrics["t"][-1] - (t_prev[-1] if t_prev else 0.0)
    assert res.hk == pytest.approx(hk_ref, rel=tol)
    if kw["time_step"] == "spectral":
        assert res.radspec == pytest.approx(st.radspec[-1], rel=tol)
    got = np.array([res.self_bias, res.self_bias_data, res.bias_data, res.bias])
    want = np.array([st.metrics[k][-1] for k in ("self-bias", "self-bias-data", "bias-data", "bias")])
    assert np.allclose(got, want, rtol=tol), (got, want)


@pytest.mark.parametrize("kind", ["flat", "clustered", "gapped", "tiny"])
def test_spectral_rule_lambda_max_by_repeated_squaring(eng_mod, kind):
    """time_step='spectral' (ces/calibrate.py:249-251: hk = 1 / max Re eig(D)): the engine takes lambda_max of the n x n
    whitened S_ee / N by 36 squarings on the matrix pipe with a two-sided bound from the Frobenius norms (relative error
    <= 2^-36 (ln n) / 4 whatever the spectrum) -- no iteration that could stall on clustered leading eigenvalues, where the
    Lanczos iteration of rounds 1-4 walked the whole Krylov space or ran into its step cap.  Against numpy's eigvalsh."""
    rng = np.random.default_rng(11)
    p, n, J = (16, 200, 1500) if kind != "tiny" else (3, 5, 64)
    d = _synthetic(p, n, J, seed=2)
    if kind == "flat":
        d["G"] = rng.standard_normal((n, J))                       # Marchenko-Pastur bulk: neighbours a fraction of a percent apart
    elif kind == "clustered":
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        s = np.ones(n); s[:4] = [3.0, 3.0 * (1 - 1e-7), 3.0 * (1 - 2e-7), 2.9999]
        d["G"] = (Q * s) @ rng.standard_normal((n, J))             # leading eigenvalues 1e-7 apart (relative)
    eng = eng_mod.Engine(p, n, J, dtype="float64")
    eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    eng.step(eng_mod.step_params(update="aldi", time_step="spectral"), d["U0"], d["G"], xi=d["xi"])
    res = eng.result()
    E = d["G"] - d["G"].mean(axis=1, keepdims=True)
    lam = np.linalg.eigvalsh(E @ E.T / J / 0.01).max()
    assert res.radspec == pytest.approx(lam, rel=1e-9)
    assert res.hk == pytest.approx(1.0 / lam, rel=1e-9)


@pytest.mark.parametrize("update,ts", [("eks", None), ("aldi", "constant")])
def test_warm_start_that_does_not_converge_falls_back_to_the_factorisation(eng_mod, monkeypatch, update, ts):
    """The sweeps of a warm start are sized from the LAST step's start (two or three inside a run).  When the ensemble then
    jumps -- another ensemble altogether, another shape of its covariance --, the previous inverse is either too far off to be tried
    or the few sweeps leave a residual above 1e-10: the closing residual launch says so and the factorisation chain runs.
    Every step of such a chain equals the always-factoring one (CESX_NS_WARM=0) to 1e-9, whichever way its inverse went."""
    p, n, J = 96, 80, 4096
    d = _synthetic(p, n, J, seed=17, dense=True)
    d["sigma"] = 1e-3 * d["sigma"]          # (a prior as tight as hk C: the EKS rule's Sigma + hk C then follows the ensemble)
    rngj = np.random.default_rng(99)
    # the jump: unrelated to the chain, its spread between a tenth and twenty times the chain's from one parameter to the next
    Uj = d["ustar"] + np.exp(rngj.uniform(np.log(0.05), np.log(10.0), (p, 1))) * rngj.standard_normal((p, J))
    kw = dict(time_step=ts, delta_t=0.02, spinup=0.0)

    def chain():
        eng = eng_mod.Engine(p, n, J, dtype="float64")
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        U, t_last, outs, warm = d["U0"], 0.0, [], []
        rng = np.random.default_rng(5)
        for i in range(7):
            if i == 4:
                U = Uj
            G = d["A"] @ U + 0.05 * np.sin(d["A"] @ U)
            xi = rng.standard_normal((p, J))
            prm = eng_mod.step_params(update=update, first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i, **kw)
            out = eng.step(prm, U, G, xi=xi).cpu().numpy().astype(np.float64)
            res = eng.result()
            warm.append(eng.warm_inverse())
            outs.append((out, res.hk, res.bias_data))
            U, t_last = U + 0.1 * (out - U), res.t_new
        return outs, warm
    monkeypatch.setenv("CESX_NS_WARM", "0")
    ref, w0 = chain()
    monkeypatch.delenv("CESX_NS_WARM")
    got, w1 = chain()
    assert w0 == [0] * 7
    assert w1[0] == 0 and w1[2:4] == [1, 1] and w1[4] == 0, w1      # cold, warm ..., the jump factors (the steps behind it either way)
    for (a, ha, ba), (b, hb, bb) in zip(got, ref):
        assert rel_err(a, b) < 1e-9 and abs(ha - hb) <= 1e-12 * abs(hb) and abs(ba - bb) <= 1e-9 * abs(bb)


@pytest.mark.parametrize("update,ts", [("eks", None), ("aldi", "constant"), ("eks", "constant"), ("aldi", "mix")])
def test_warm_started_spd_inverses_equal_the_factored_ones(eng_mod, monkeypatch, update, ts):
    """The hk-dependent SPD inverses of K2 ((Sigma + hk C)^-1 of the EKS rule, (hk C_gg + Gamma)^-1 of the recomputed gain) start
    from the previous step's inverse inside a chain (Newton-Schulz sweeps, the true residual checked, the factorisation chain
    skipped only then).  A chain on one engine: the first step factors (cold), the later ones take the warm start -- and the
    ensembles and scalars equal those of a chain that always factors (CESX_NS_WARM=0) to 1e-9."""
    from oracle import ces_numpy as oc
    p, n, J = 96, 80, 4096
    d = _synthetic(p, n, J, seed=13, dense=True)
    kw = dict(time_step=ts, delta_t=0.02, spinup=0.0)

    def chain():
        eng = eng_mod.Engine(p, n, J, dtype="float64")
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        U, t_last, outs, warm = d["U0"], 0.0, [], []
        rng = np.random.default_rng(5)
        for i in range(4):
            G = d["A"] @ U + 0.05 * np.sin(d["A"] @ U)
            xi = rng.standard_normal((p, J))
            prm = eng_mod.step_params(update=update, first_step=(i == 0), t_len=min(i, 1), t_last=t_last + 2.0 * (ts == "mix"),
                                      step_index=i, **kw)
            out = eng.step(prm, U, G, xi=xi).cpu().numpy().astype(np.float64)
            res = eng.result()
            warm.append(eng.warm_inverse())
            outs.append((out, res.hk, res.bias_data))
            # (small moves: the next ensemble is a convex step towards the update, as late in a run)
            U, t_last = U + 0.1 * (out - U), res.t_new
        return outs, warm
    monkeypatch.setenv("CESX_NS_WARM", "0")
    ref, w0 = chain()
    monkeypatch.delenv("CESX_NS_WARM")
    got, w1 = chain()
    assert w0 == [0, 0, 0, 0]
    # (time_step='mix': its first step takes the Frobenius step size, the second delta_t -- another matrix, factored once more)
    assert w1[0] == 0 and w1[-2:] == [1, 1] and (ts == "mix" or w1[1] == 1), w1
    for (a, ha, ba), (b, hb, bb) in zip(got, ref):
        assert rel_err(a, b) < 1e-9 and abs(ha - hb) <= 1e-12 * abs(hb) and abs(ba - bb) <= 1e-9 * abs(bb)
    # ... and the first (cold) step is the pinned oracle's
    st = oc.OracleState(p, n, J, d["mu"], d["sigma"], d["ustar"])
    G0 = d["A"] @ d["U0"] + 0.05 * np.sin(d["A"] @ d["U0"])
    xi0 = np.random.default_rng(5).standard_normal((p, J))
    okw = {k: v for k, v in kw.items() if v is not None and (k != "spinup" or ts == "mix")}
    if ts != "mix":
        ref0 = oc.factored_step(st, d["y"], d["U0"], G0, d["Gamma"], xi0, update=update, **okw)
        assert rel_err(got[0][0], ref0) < TOL64


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_moments_against_oracle(eng_mod, dtype):
    """K1 alone: ubar, gbar, C, K, M from the engine vs. the oracle's moments."""
    from oracle import ces_numpy as oc
    p, n, J = 48, 40, 3000
    d = _synthetic(p, n, J, seed=5, dense=True)
    eng = eng_mod.Engine(p, n, J, dtype=dtype)
    eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    eng.step(eng_mod.step_params(update="aldi"), d["U0"], d["G"], xi=d["xi"])
    eng.result()
    dd = eng.debug_dense()
    cast = (lambda a: a.astype(np.float32).astype(np.float64)) if dtype == "float32" else (lambda a: a)
    ubar, gbar, S_uu, S_ug, S_ee, _, _ = oc.moments(cast(d["U0"]), cast(d["G"]))
    tol = 1e-10 if dtype == "float64" else 2e-5
    assert rel_err(dd["ubar"], ubar.ravel()) < tol and rel_err(dd["gbar"], gbar.ravel()) < tol
    C = S_uu / (J - 1) + 1e-8 * np.eye(p)
    assert rel_err(dd["C"], C) < tol
    assert rel_err(dd["L"], np.linalg.cholesky(C)) < tol * 50
    assert rel_err(dd["K"], np.linalg.solve(d["Gamma"].T, (S_ug / J).T).T) < tol * 50
    assert rel_err(dd["M"], np.linalg.solve(d["sigma"].T, C.T).T) < tol * 50


@pytest.mark.parametrize("dtype,tol", [("float32", 2e-5), ("float64", 1e-12)])
def test_device_noise_matches_philox_oracle(eng_mod, dtype, tol):
    from oracle import philox
    p, n, J = 7, 3, 1000
    eng = eng_mod.Engine(p, n, J, dtype=dtype, seed=0x1234567890, j_offset=0)
    xi = eng.draw_noise(5).cpu().numpy().astype(np.float64)
    want = philox.noise_block(p, J, 0x1234567890, 5, dtype=dtype)
    assert np.max(np.abs(xi - want)) < tol * 10
    assert abs(xi.mean()) < 0.05 and abs(xi.std() - 1.0) < 0.05


def test_noise_does_not_depend_on_sharding(eng_mod):
    """Philox counters use the GLOBAL particle index."""
    p, n, J = 6, 3, 512
    whole = eng_mod.Engine(p, n, J, dtype="float32", seed=9).draw_noise(2).cpu().numpy()
    a = eng_mod.Engine(p, n, 200, dtype="float32", seed=9, J_global=J, j_offset=0).draw_noise(2).cpu().numpy()
    b = eng_mod.Engine(p, n, 312, dtype="float32", seed=9, J_global=J, j_offset=200).draw_noise(2).cpu().numpy()
    assert np.array_equal(np.concatenate([a, b], axis=1), whole)


def test_in_kernel_noise_equals_injected_noise(eng_mod):
    """xi drawn inside the update kernel == the same block injected from memory."""
    p, n, J = 40, 24, 1500
    d = _synthetic(p, n, J, seed=3)
    eng = eng_mod.Engine(p, n, J, dtype="float32", seed=77)
    eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    prm = eng_mod.step_params(update="aldi", step_index=4)
    a = eng.step(prm, d["U0"], d["G"], xi=None).cpu().numpy()
    eng.result()
    xi = eng.draw_noise(4)
    b = eng.step(prm, d["U0"], d["G"], xi=xi).cpu().numpy()
    eng.result()
    assert rel_err(a, b) < 1e-6


@pytest.mark.parametrize("dtype,tol", [("float32", 2e-6), ("float64", 1e-13)])
@pytest.mark.parametrize("update", ["aldi", "aldi_constant"])
def test_prefetched_noise_block(eng_mod, dtype, tol, update):
    """cesx_prefetch_noise (include/cesx.h): the block drawn AHEAD of the update on the side stream is what the
    update uses when xi_dev == NULL and the step index matches -- the same numbers as the in-kernel generator
    (CESX_NO_NOISE_PREFETCH=1) and as the injected cesx_draw_noise block; a different step index falls back to
    drawing inside the update kernel.  Split entry points (what ces_amd.dist drives) and cesx_step."""
    import os
    from ces_amd.dist import ShardedUpdate
    p, n, J = 48, 40, 2048
    d = _synthetic(p, n, J, seed=13)

    def run(prefetch_step, use_step, env=None):
        if env:
            os.environ["CESX_NO_NOISE_PREFETCH"] = "1"
        try:
            eng = eng_mod.Engine(p, n, J, dtype=dtype, seed=21)
            eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
            sh = ShardedUpdate(eng)
            U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
            prm = eng_mod.step_params(update=update, step_index=use_step)
            sh.begin(prm, U, G, recenter=True, noise_step=prefetch_step)
            out = sh.finish(prm, U, G, xi=None)
            res = sh.result()
            return out.cpu().numpy(), res.hk, eng
        finally:
            os.environ.pop("CESX_NO_NOISE_PREFETCH", None)
    a, hk_a, eng = run(7, 7)                     # prefetched block used
    b, hk_b, _ = run(None, 7)                    # no prefetch: drawn inside the update kernel
    c, hk_c, _ = run(7, 7, env=True)             # prefetch disabled by the environment
    e, hk_e, _ = run(3, 7)                       # block of another step prefetched: ignored, step 7 drawn in-kernel
    assert rel_err(a, b) < tol and rel_err(a, c) < tol and rel_err(a, e) < tol
    assert hk_a == pytest.approx(hk_b, rel=1e-12) and hk_a == pytest.approx(hk_e, rel=1e-12)
    # and the injected block of the same step gives the same update
    xi = eng.draw_noise(7)
    prm = eng_mod.step_params(update=update, step_index=7)
    f = eng.step(prm, d["U0"], d["G"], xi=xi).cpu().numpy()
    eng.result()
    assert rel_err(a, f) < tol
    # a different step index draws a different block
    g, _, _ = run(8, 8)
    assert rel_err(a, g) > 1e-3


def test_noise_lookahead_with_irregular_step_indices(eng_mod, monkeypatch):
    """The two-buffer lookahead (the block of step s + 1 drawn behind chol(C) of step s) must never hand an update a
    block of the wrong step: chains whose step indices jump, repeat and go backwards give, step by step, the same
    ensembles as the same chains with the lookahead off and with the noise drawn inside the update kernel."""
    from ces_amd.dist import ShardedUpdate
    p, n, J = 64, 48, 4096
    d = _synthetic(p, n, J, seed=91)
    steps = [5, 6, 7, 9, 3, 4, 4, 5, 11]
    outs = []
    for mode in ("lookahead", "single", "in_kernel"):
        monkeypatch.setenv("CESX_NOISE_LOOKAHEAD", "0" if mode == "single" else "1")
        if mode == "in_kernel":
            monkeypatch.setenv("CESX_NO_NOISE_PREFETCH", "1")
        else:
            monkeypatch.delenv("CESX_NO_NOISE_PREFETCH", raising=False)
        eng = eng_mod.Engine(p, n, J, dtype="float32", seed=5)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        sh = ShardedUpdate(eng)
        U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
        bufs = [eng.empty(p), eng.empty(p)]
        chain = []
        for k, si in enumerate(steps):
            prm = eng_mod.step_params(update="aldi", first_step=(k == 0), t_len=min(k, 1), t_last=0.01 * k, step_index=si)
            sh.begin(prm, U, G, recenter=(k == 0), noise_step=si)
            U = sh.finish(prm, U, G, xi=None, out=bufs[k % 2])
            chain.append(U.cpu().numpy().copy())
            sh.result()
        outs.append(chain)
    for k in range(len(steps)):
        assert np.array_equal(outs[0][k], outs[1][k]), "lookahead vs single buffer, step %d" % k
        assert rel_err(outs[0][k], outs[2][k]) < 2e-6, "prefetched vs in-kernel noise, step %d" % k


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 1e-4)])
def test_logical_shards_add_up(eng_mod, dtype, tol):
    """moments of N column shards sum to the moments of the whole
    ensemble, and apply() on each shard reproduces the single-device step."""
    import torch
    p, n, J = 24, 20, 3000
    d = _synthetic(p, n, J, seed=8)
    whole = eng_mod.Engine(p, n, J, dtype=dtype)
    whole.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    prm = eng_mod.step_params(update="aldi")
    ref = whole.step(prm, d["U0"], d["G"], xi=d["xi"]).cpu().numpy()
    whole.result()
    cuts = [0, 1000, 1800, J]
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        e = eng_mod.Engine(p, n, b - a, dtype=dtype, J_global=J, j_offset=a)
        e.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        shards.append((e, e.to_device(d["U0"][:, a:b]), e.to_device(d["G"][:, a:b]), e.to_device(d["xi"][:, a:b])))
    sums = sum(e.colsum(U, G) for e, U, G, _ in shards)           # stand-in for the all-reduce
    for e, *_ in shards:
        e.set_shift(sums)
    mom = sum(e.moments(U, G) for e, U, G, _ in shards)
    outs = [e.apply(prm, mom, U, G, xi=xi).cpu().numpy() for e, U, G, xi in shards]
    for e, *_ in shards:
        e.result()
    torch.cuda.synchronize()
    assert rel_err(np.concatenate(outs, axis=1), ref) < tol


def test_forward_lineal_hook(eng_mod):
    """ces/utils.py:25-31 on the whole shard: the one-call form (cesx_forward_lineal) and the
    installed-map form (cesx_forward_set_lineal + cesx_forward_apply, what lineal.forward_device uses), through the
    LDS-DMA kernels (J % 4 == 0) and the register-staged fallback (ragged J), two models sharing one engine."""
    from ces_amd.utils import lineal
    rng = np.random.default_rng(2)
    for p, n, J in ((37, 21, 1234), (64, 50, 2048), (300, 260, 1024)):
        A, b, U = rng.standard_normal((n, p)), rng.standard_normal(n), rng.standard_normal((p, J))
        A2 = rng.standard_normal((n, p))
        for dtype, tol in (("float64", 1e-12), ("float32", 1e-5)):
            eng = eng_mod.Engine(p, n, J, dtype=dtype)
            Ud = eng.to_device(U)
            m1, m2 = lineal(A, b=b), lineal(A2)
            assert rel_err(m1.forward_device(eng, Ud).cpu().numpy(), A @ U + b[:, None]) < tol
            assert rel_err(m2.forward_device(eng, Ud).cpu().numpy(), A2 @ U) < tol
            assert rel_err(m1.forward_device(eng, Ud).cpu().numpy(), A @ U + b[:, None]) < tol       # re-installed
            assert rel_err(eng.forward_lineal(A2, Ud, b=b).cpu().numpy(), A2 @ U + b[:, None]) < tol


def test_c4_darcy_drop_in(eng_mod):
    """BASELINE.json configs[3] shape: Darcy forward map on the host (ces_amd/darcy.py,
    model_trunc(p=64), 50 observations) + ensemble update on the GPU, driven exactly like
    examples/scripts/darcy-flow.py:43-93; every step is checked against the literal oracle."""
    from ces_amd.calibrate import sampling
    from ces_amd import darcy
    from oracle import ces_numpy as oc
    full = darcy.model(); full.set_initial(); full.n_obs = 50
    Ufull = full(full.ustar, full_solution=True)
    np.random.seed(1)
    obs_index = np.random.choice(int(full.p), 50, replace=False, p=Ufull / Ufull.sum())
    model = darcy.model_trunc(p=64); model.set_initial(); model.n_obs = 50; model.obs_index = obs_index
    gamma = 0.005
    Gamma = gamma ** 2 * np.identity(50)
    y_obs = model(model.ustar) + gamma * np.random.normal(0, 1, 50)
    J = 192
    eks = sampling(p=model.p, n_obs=model.n_obs, J=J)
    eks.ustar = model.ustar.reshape(model.p, -1)
    eks.T = 3
    eks.mu = np.zeros((model.p, 1)); eks.sigma = 100. * np.identity(model.p)
    np.random.seed(0)
    U0 = 10 * np.random.normal(0, 1, [eks.p, J])
    np.random.seed(5)
    eks.run(y_obs, U0, model, Gamma, np.linalg.cholesky(Gamma), t_tol=5)
    assert eks.Uall.shape == (4, 64, J) and eks.Gall.shape == (4, 50, J)
    np.random.seed(5)
    st = oc.OracleState(64, 50, J, eks.mu, eks.sigma, eks.ustar, T=3)
    for i in range(3):
        st.trace_len = i + 1
        xi = np.random.normal(0, 1, [64, J])
        ref = oc.literal_step(st, y_obs, eks.Uall[i], eks.Gall[i], Gamma, xi, update="aldi")
        assert rel_err(eks.Uall[i + 1], ref) < TOL64
    assert np.allclose(eks.metrics["t"], st.metrics["t"], rtol=1e-8)
    assert np.allclose(eks.metrics["bias-data"], st.metrics["bias-data"], rtol=1e-7)


def test_save_load_round_trip(eng_mod, tmp_path):
    """On-disk format of enka.save / enka.load (ces/calibrate.py:170-237): file names,
    metrics pickle, path arrays, per-iteration online dumps."""
    import os
    from ces_amd.calibrate import sampling
    from ces_amd.utils import lineal
    rng = np.random.default_rng(4)
    p, n, J = 3, 5, 30
    A = rng.standard_normal((n, p))
    model = lineal(A); model.l_window = 7                      # run(save_online=True) reads model.l_window (:376)
    eks = sampling(p=p, n_obs=n, J=J); eks.T = 4
    eks.ustar = np.ones((p, 1)); eks.mu = np.zeros((p, 1)); eks.sigma = 10.0 * np.eye(p)
    eks.directory = str(tmp_path); eks.nexp = 3
    y = A @ np.ones(p)
    eks.run(y, rng.standard_normal((p, J)), model, 0.1 * np.eye(n), None, save_online=True, t_tol=1e9)
    d = tmp_path / "ensembles" / "lineal-eks-007-0030-03"
    assert sorted(os.listdir(d)) == ["Gensemble_000%d.npy" % i for i in range(4)] + \
        ["ensemble_000%d.npy" % i for i in range(4)] + ["metrics.pkl"]
    assert eks.online_path.endswith("/ensembles/lineal-0030-03/")
    eks.save(path=str(tmp_path) + "/", file="final/", all=True)
    assert sorted(os.listdir(tmp_path / "final")) == ["Gensemble.npy", "Gensemble_path.npy", "ensemble.npy",
                                                      "ensemble_path.npy", "metrics.pkl"]
    other = sampling(p=p, n_obs=n, J=J)
    assert other.load(path=str(tmp_path) + "/", eks_dir="final/")
    assert np.array_equal(other.Uall, eks.Uall) and other.metrics["t"] == eks.metrics["t"]
    third = sampling(p=p, n_obs=n, J=1)
    assert third.load(path=str(d) + "/", eks_dir="", ix_ensemble=True)
    assert third.Uall.shape == (4, p, J) and third.J == J and np.array_equal(third.Ustar, eks.Uall[3])
    # resume: a second run on the same object appends (ces/calibrate.py:307-310)
    n_before = len(eks.metrics["t"])
    eks.run(y, eks.Ustar, model, 0.1 * np.eye(n), None, t_tol=1e9)
    assert len(eks.metrics["t"]) == n_before + 4 and eks.Uall.shape[0] == 10
    assert eks.metrics["t"][n_before] > eks.metrics["t"][n_before - 1]      # pseudo-time keeps accumulating


def test_overlap_variants_match(eng_mod, monkeypatch):
    """CESX_OVERLAP=0 (Cholesky in line instead of beside the second Gram launch) gives the same step."""
    p, n, J = 96, 80, 5000
    d = _synthetic(p, n, J, seed=12)
    outs = []
    for ov in ("1", "0"):
        monkeypatch.setenv("CESX_OVERLAP", ov)
        eng = eng_mod.Engine(p, n, J, dtype="float32", seed=3)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        out = eng.step(eng_mod.step_params(update="aldi", step_index=2), d["U0"], d["G"], xi=None)
        res = eng.result()
        outs.append((out.cpu().numpy(), res.hk, res.bias_data, res.self_bias_data))
    assert rel_err(outs[1][0], outs[0][0]) < 1e-5
    assert outs[1][1:] == pytest.approx(outs[0][1:], rel=1e-6)


@pytest.mark.parametrize("dtype,tol", [("float32", 1e-5), ("float64", 1e-12)])
@pytest.mark.parametrize("update", ["aldi", "eks", "aldi_constant"])
@pytest.mark.parametrize("shape", [(96, 80, 5000), (300, 40, 1004), (33, 17, 260)])
def test_update_kernels_v1_v2_match(eng_mod, monkeypatch, update, shape, dtype, tol):
    """K3 through the LDS-DMA kernels (kernels_update2.hip fp32 / kernels_update3.hip fp64, default) and through
    the register-staged one (CESX_UPDATE_V1=1) give the same step: same Philox noise, same data
    metrics, ragged J / p / n included (p > 256 takes two row chunks)."""
    p, n, J = shape
    d = _synthetic(p, n, J, seed=21)
    outs = []
    for v1 in ("0", "1"):
        monkeypatch.setenv("CESX_UPDATE_V1", v1)
        eng = eng_mod.Engine(p, n, J, dtype=dtype, seed=5)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        out = eng.step(eng_mod.step_params(update=update, step_index=4, first_step=False, t_len=1, t_last=0.1),
                       d["U0"], d["G"], xi=None)
        res = eng.result()
        outs.append((out.cpu().numpy(), res.hk, res.bias_data, res.self_bias_data))
    assert rel_err(outs[1][0], outs[0][0]) < tol
    assert outs[1][1:] == pytest.approx(outs[0][1:], rel=tol)


@pytest.mark.parametrize("dtype,tol", [("float32", 1e-5), ("float64", 1e-12)])
@pytest.mark.parametrize("update,prefetch", [("aldi", True), ("aldi", False), ("eks", True), ("aldi_constant", True)])
@pytest.mark.parametrize("shape", [(64, 50, 8192), (40, 24, 200), (16, 150, 1000), (33, 17, 260), (64, 64, 512)])
def test_small_update_kernels_match_the_tiled_ones(eng_mod, monkeypatch, update, prefetch, shape, dtype, tol):
    """K3 for coefficient matrices of at most 64 rows and 192 columns (update2s_kernel / update3s_kernel: the workgroup's
    whole tile LDS resident -- the reference's own problem sizes, examples/scripts/darcy-flow.py:97-105) against the tiled
    LDS-DMA kernels (CESX_UPDATE_SMALL=0) on the same step: ragged J (not a multiple of the 64 / 32 particles of a
    workgroup), p < 64, the widest G segment that still qualifies (n = 150), the largest image that does (64 x 192), the noise block read from memory and
    (fp32) drawn in the kernel, the hk-free and the assembled coefficient images, both passes of `aldi_constant`."""
    p, n, J = shape
    d = _synthetic(p, n, J, seed=23)
    if not prefetch:
        monkeypatch.setenv("CESX_NO_NOISE_PREFETCH", "1")
    outs = []
    for small in ("1", "0"):
        monkeypatch.setenv("CESX_UPDATE_SMALL", small)
        eng = eng_mod.Engine(p, n, J, dtype=dtype, seed=5)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        U, chain = eng.to_device(d["U0"]), []
        G = eng.to_device(d["G"])
        t_last = 0.0
        for i in range(2):
            prm = eng_mod.step_params(update=update, step_index=4 + i, first_step=(i == 0), t_len=i, t_last=t_last)
            U = eng.step(prm, U, G, xi=None, recenter=(i == 0))
            res = eng.result()
            t_last = res.t_new
            chain.append((res.hk, res.bias_data, res.self_bias_data))
        outs.append((U.cpu().numpy(), chain))
    assert rel_err(outs[1][0], outs[0][0]) < 20 * tol          # (two steps: the second one starts from the first one's rounding)
    for a, b in zip(outs[0][1], outs[1][1]):
        assert a == pytest.approx(b, rel=20 * tol)


def test_comm_overlap_stream_path_matches(eng_mod, monkeypatch):
    """ShardedUpdate with the head all-reduce + chol(C) on a second stream (the multi-GPU path,
    forced here on one rank) gives bit-identical steps to the in-order path."""
    from ces_amd.dist import ShardedUpdate
    p, n, J = 128, 96, 8192
    d = _synthetic(p, n, J, seed=31)
    outs = []
    for ov in (False, True):
        eng = eng_mod.Engine(p, n, J, dtype="float32", seed=9)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        sh = ShardedUpdate(eng, overlap_comm=ov)
        U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
        t_last, chain = 0.0, []
        for i in range(4):
            prm = eng_mod.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i)
            U = sh.step(prm, U, G, xi=None, recenter=(i == 0))
            res = sh.result()
            t_last = res.t_new
            chain.append((res.hk, res.t_new, res.bias_data))
        outs.append((U.cpu().numpy(), chain))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_single_device_fast_path_variants_match(eng_mod, monkeypatch, dtype):
    """The single-device fast path -- hand-over events bound to kernels (cesx_moments_uu_chol), the metric finalisation +
    publication riding on the NEXT step's U x U reduce launch, the noise block drawn one step ahead (CESX_NOISE_LOOKAHEAD) or
    for the current step only -- gives bit-identical chains in a pipelined loop (begin(i+1) before result(i)) and in a
    step-by-step one (where every result read flushes the deferred publication as a kernel of its own).  (Round 6 removed the
    CESX_EXT_EVENTS=0 / CESX_DEFER_PUBLISH=0 arms this test also used to run: no caller took them.)"""
    from ces_amd.dist import ShardedUpdate
    p, n, J = 128, 96, 8192
    d = _synthetic(p, n, J, seed=77)
    outs = []
    # (the assembled form of the coefficient matrix on both sides: the hk-free form of the fused launch rounds differently,
    #  test_hk_free_update_matches_the_assembled_form holds it to the assembled one)
    monkeypatch.setenv("CESX_HKFREE", "0")
    for fast, pipelined in ((True, True), (True, False), (False, True), (False, False)):
        monkeypatch.setenv("CESX_NOISE_LOOKAHEAD", "1" if fast else "0")
        eng = eng_mod.Engine(p, n, J, dtype=dtype, seed=9)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        sh = ShardedUpdate(eng)
        U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
        bufs = [eng.empty(p), eng.empty(p)]
        t_last, chain = 0.0, []

        def prm_of(i, t_last):
            return eng_mod.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i)
        nsteps = 5
        sh.begin(prm_of(0, 0.0), U, G, recenter=True, noise_step=0)
        for i in range(nsteps):
            out = sh.finish(prm_of(i, t_last), U, G, xi=None, out=bufs[i % 2])
            if pipelined and i + 1 < nsteps:
                sh.begin(prm_of(i + 1, 0.0), out, G, noise_step=i + 1)      # (begin reads prm.update only)
            res = sh.result()
            if not pipelined and i + 1 < nsteps:
                sh.begin(prm_of(i + 1, 0.0), out, G, noise_step=i + 1)
            t_last = res.t_new
            chain.append((res.hk, res.t_new, res.bias, res.self_bias, res.bias_data, res.self_bias_data,
                          res.lag_bias_data, res.lag_self_bias_data))
            U = out
        outs.append((U.cpu().numpy().copy(), chain))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0])
        assert o[1] == outs[0][1]


@pytest.mark.parametrize("dtype,p,n,J", [("float32", 256, 256, 16384), ("float32", 96, 80, 4096), ("float64", 128, 96, 4096),
                                         ("float64", 250, 130, 8192)])
def test_polled_side_stream_join_is_bit_identical(eng_mod, monkeypatch, dtype, p, n, J):
    """One device, ALDI, default time step, diagonal Gamma / Sigma: the caller's stream joins the side stream
    (U-only centring -> chol(C)) through the word the factorisation stores last -- workgroup 0 of the G-part centring
    launch polls it, the assembly launch reads that stream's results with agent-scope loads -- instead of a barrier
    packet (CESX_POLL_JOIN=0).  Same arithmetic on the same numbers: bit-identical chains, pipelined (the
    factorisation finishes beside the second Gram launch) and step by step, at small J (the caller's stream reaches the
    join long before the factorisation ends) and with injected noise (no draw behind the factorisation)."""
    from ces_amd.dist import ShardedUpdate
    d = _synthetic(p, n, J, seed=p + J)
    outs = []
    for poll, pipelined, inject in (("1", True, False), ("0", True, False), ("1", False, False), ("1", True, True), ("0", True, True)):
        monkeypatch.setenv("CESX_POLL_JOIN", poll)
        eng = eng_mod.Engine(p, n, J, dtype=dtype, seed=11)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        sh = ShardedUpdate(eng)
        U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
        xi = eng.to_device(d["xi"]) if inject else None
        bufs = [eng.empty(p), eng.empty(p)]
        t_last, chain = 0.0, []

        def prm_of(i, t_last):
            return eng_mod.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i)
        nsteps = 6
        sh.begin(prm_of(0, 0.0), U, G, recenter=True, noise_step=None if inject else 0)
        for i in range(nsteps):
            out = sh.finish(prm_of(i, t_last), U, G, xi=xi, out=bufs[i % 2])
            if pipelined and i + 1 < nsteps:
                sh.begin(prm_of(i + 1, 0.0), out, G, noise_step=None if inject else i + 1)
            res = sh.result()
            if not pipelined and i + 1 < nsteps:
                sh.begin(prm_of(i + 1, 0.0), out, G, noise_step=None if inject else i + 1)
            t_last = res.t_new
            chain.append((res.hk, res.t_new, res.bias, res.self_bias, res.bias_data, res.self_bias_data))
            U = out
        outs.append((U.cpu().numpy().copy(), chain))
    assert np.isfinite(outs[0][0]).all()
    for k in (1, 2):
        assert np.array_equal(outs[k][0], outs[0][0]) and outs[k][1] == outs[0][1]
    assert np.array_equal(outs[4][0], outs[3][0]) and outs[4][1] == outs[3][1]


def _aldi_chain(eng_mod, d, p, n, J, dtype, nsteps=4, pipelined=False, stream=None):
    """An ALDI chain on one engine through ShardedUpdate.begin / finish (as bench.py and ShardedSampler drive it).
    Returns (engine, last ensemble, per-step scalars, the driver)."""
    import torch
    from ces_amd.dist import ShardedUpdate
    ctx = torch.cuda.stream(stream) if stream is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        eng = eng_mod.Engine(p, n, J, dtype=dtype, seed=9)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        sh = ShardedUpdate(eng)
        U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
        bufs = [eng.empty(p), eng.empty(p)]
        t_last, chain = 0.0, []

        def prm_of(i, t_last):
            return eng_mod.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i)
        sh.begin(prm_of(0, 0.0), U, G, recenter=True, noise_step=0)
        for i in range(nsteps):
            out = sh.finish(prm_of(i, t_last), U, G, xi=None, out=bufs[i % 2])
            if pipelined and i + 1 < nsteps:
                sh.begin(prm_of(i + 1, 0.0), out, G, noise_step=i + 1)
            res = sh.result()
            if not pipelined and i + 1 < nsteps:
                sh.begin(prm_of(i + 1, 0.0), out, G, noise_step=i + 1)
            t_last = res.t_new
            chain.append((res.hk, res.t_new, res.bias, res.self_bias, res.bias_data, res.self_bias_data))
            U = out
        torch.cuda.synchronize()
        return eng, U.cpu().numpy().copy(), np.array(chain), sh
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)


@pytest.mark.parametrize("p,n,J", [(256, 256, 16384), (128, 96, 8192), (96, 80, 4096), (40, 24, 2048)])
def test_hk_free_update_matches_the_assembled_form(eng_mod, monkeypatch, p, n, J):
    """fp32, ALDI, default time step, diagonal Gamma / Sigma: K2 keeps the time step OUT of the coefficient matrix
    (tail_aldi_kernel: W = [L | a I - M + I/hk | -K], the factorisation stores L into the image itself) and K3 applies hk
    and sqrt(2 hk) at run time (xi segment first, one rescale, result times hk).  Same numbers up to fp32 rounding as
    the assembled form W = [(1 + hk a) I - hk M | -hk K | sqrt(2hk) L] (CESX_HKFREE=0), with bit-identical step sizes as
    long as the ensembles agree (first step) and deterministic from run to run; pipelined and step by step, polled and
    event-joined."""
    d = _synthetic(p, n, J, seed=p + n)
    monkeypatch.setenv("CESX_HKFREE", "0")
    _, U0, c0, _ = _aldi_chain(eng_mod, d, p, n, J, "float32")
    monkeypatch.setenv("CESX_HKFREE", "1")
    _, U1, c1, _ = _aldi_chain(eng_mod, d, p, n, J, "float32")
    _, U2, c2, _ = _aldi_chain(eng_mod, d, p, n, J, "float32", pipelined=True)
    monkeypatch.setenv("CESX_POLL_JOIN", "0")
    _, U3, c3, _ = _aldi_chain(eng_mod, d, p, n, J, "float32", pipelined=True)
    assert np.array_equal(c1[0, :2], c0[0, :2])                      # hk, t of the first step: the same sums in the same order
    assert np.allclose(c1[0, 2:4], c0[0, 2:4], rtol=1e-12, atol=0)   # bias, self-bias: the tail launch sums the trace row by row
    scale = np.max(np.abs(U0))
    assert np.max(np.abs(U1 - U0)) <= 2e-5 * scale, np.max(np.abs(U1 - U0)) / scale
    assert np.allclose(c1, c0, rtol=2e-5, atol=0)
    for U, c in ((U2, c2), (U3, c3)):                                 # the same launches on the same numbers
        assert np.array_equal(U, U1) and np.array_equal(c, c1)


@pytest.mark.parametrize("p,n,J", [(256, 256, 8192), (250, 100, 1000), (233, 37, 4100)])
def test_update_through_the_cholesky_factor(eng_mod, monkeypatch, p, n, J):
    """K3 through the Cholesky factor (kernels_update4.hip; fp32, diagonal Sigma, 224 < p <= 256): C Sigma^{-1} (U - mu) =
    L (L^T Sigma^{-1} U) - C Sigma^{-1} mu with C = L L^T as factored (ces/calibrate.py:476-478, :484-488).  Against the pinned
    oracle with an injected block (ragged p, n and J included: the last 32-row block, the last G tile and the last workgroup
    are partial), against the dense hk-free form (CESX_CHAIN=0) to fp32 rounding, a non-zero prior mean (the C Sigma^{-1} mu
    term lives in the bias), the factor cesx_debug_dense reports after a step that kept it in the image only, and the block
    the engine draws itself when none was injected or drawn ahead."""
    from oracle import ces_numpy as oc
    d = _synthetic(p, n, J, seed=p + n + J + 1)
    rng = np.random.default_rng(5)
    mu = 0.3 * rng.standard_normal((p, 1))
    sigma = np.diag(100.0 * (1.0 + 0.5 * rng.random(p)))
    st = oc.OracleState(p, n, J, mu, sigma, d["ustar"])
    ref = oc.factored_step(st, d["y"], d["U0"], d["G"], d["Gamma"], d["xi"], update="aldi")
    outs = {}
    for chain in ("1", "0"):
        monkeypatch.setenv("CESX_CHAIN", chain)
        eng = eng_mod.Engine(p, n, J, dtype="float32", seed=3)
        eng.set_problem(d["y"], d["Gamma"], mu, sigma, d["ustar"])
        out = eng.step(eng_mod.step_params(update="aldi", step_index=4), d["U0"], d["G"], xi=d["xi"])
        res = eng.result()
        assert eng.update_form() == (2 if chain == "1" else 1)
        outs[chain] = out.cpu().numpy()
        assert rel_err(outs[chain], ref) < TOL32
        assert res.hk == pytest.approx(st.metrics["t"][-1], rel=TOL32)
        got = np.array([res.self_bias, res.self_bias_data, res.bias_data, res.bias])
        want = np.array([st.metrics[k][-1] for k in ("self-bias", "self-bias-data", "bias-data", "bias")])
        assert np.allclose(got, want, rtol=TOL32), (got, want)
        if chain == "1":
            dd = eng.debug_dense()            # (the chained step kept L in its image: re-factored on demand)
            Lref = np.linalg.cholesky(dd["C"])
            assert np.max(np.abs(dd["L"] - Lref)) <= 1e-9 * np.max(np.abs(Lref))
            # no block injected, none drawn ahead: the engine draws the step's block itself -- the same block cesx_draw_noise gives
            prm = eng_mod.step_params(update="aldi", step_index=9)
            a = eng.step(prm, d["U0"], d["G"], xi=None).cpu().numpy()
            eng.result()
            b = eng.step(prm, d["U0"], d["G"], xi=eng.draw_noise(9)).cpu().numpy()
            eng.result()
            assert eng.update_form() == 2 and np.array_equal(a, b)
    scale = np.max(np.abs(outs["0"]))
    assert np.max(np.abs(outs["1"] - outs["0"])) <= 2e-5 * scale


def test_a_step_that_turns_out_not_to_be_chained_refactors(eng_mod, monkeypatch):
    """The factorisation is enqueued (cesx_chol_async, beside the second Gram launch) before the step's time-step rule is
    known; expecting a chained step it keeps L in the coefficient image only.  When the step then takes another rule
    (here: `spectral` and `constant` in the middle of a default-rule chain, pipelined) the assembled form needs the fp64
    factor: launch_dense re-factors C in line (Engine::L_stale).  Same chain as an engine that never chains."""
    from ces_amd.dist import ShardedUpdate
    p, n, J = 256, 96, 4096
    d = _synthetic(p, n, J, seed=77)
    rules = [None, None, "spectral", None, "constant", None]

    def chain():
        eng = eng_mod.Engine(p, n, J, dtype="float32", seed=9)
        eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
        sh = ShardedUpdate(eng)
        U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
        bufs = [eng.empty(p), eng.empty(p)]
        t_last, outs, forms = 0.0, [], []
        prm0 = eng_mod.step_params(update="aldi")
        sh.begin(prm0, U, G, recenter=True, noise_step=0)
        for i, ts in enumerate(rules):
            prm = eng_mod.step_params(update="aldi", time_step=ts, delta_t=0.01, first_step=(i == 0), t_len=min(i, 1),
                                      t_last=t_last, step_index=i)
            out = sh.finish(prm, U, G, xi=None, out=bufs[i % 2])
            if i + 1 < len(rules):
                sh.begin(prm0, out, G, noise_step=i + 1)
            res = sh.result()
            forms.append(eng.update_form())
            t_last = res.t_new
            outs.append((out.cpu().numpy().copy(), res.hk, res.t_new))
            U = out
        return outs, forms
    monkeypatch.setenv("CESX_CHAIN", "1")
    a, fa = chain()
    monkeypatch.setenv("CESX_CHAIN", "0")
    b, fb = chain()
    assert fa == [2, 2, 0, 2, 0, 2] and fb == [1, 1, 0, 1, 0, 1]
    for (ua, hka, ta), (ub, hkb, tb) in zip(a, b):
        assert hka == pytest.approx(hkb, rel=1e-4) and ta == pytest.approx(tb, rel=1e-4)
        assert np.max(np.abs(ua - ub)) <= 1e-4 * np.max(np.abs(ub))


@pytest.mark.parametrize("p,n,J", [(128, 96, 8192), (256, 96, 4096)])      # (the second: K3 through the Cholesky factor, the image-only factorisation)
def test_polled_join_that_runs_out_leaves_the_step_untouched_and_is_rerun(eng_mod, monkeypatch, p, n, J):
    """The polled join of the side stream (launch_dense) is bounded in wall time.  A factorisation that never stores its
    word (CESX_TEST_DROP_CHOL_SIGNAL: the second one) makes the poll of that step run out: the assembly and update
    launches write nothing, cesx_result switches the engine to the event join and re-runs the step with chol(C) in line
    -- the chain is bit-identical to one that never polled (CESX_POLL_JOIN=0); a pipelined driver is told (CESX_ESTATE)
    that the moments it enqueued behind the failed step must be redone, which ShardedUpdate.result does by itself."""
    d = _synthetic(p, n, J, seed=81)
    monkeypatch.setenv("CESX_POLL_JOIN", "0")
    e0, U0, c0, _ = _aldi_chain(eng_mod, d, p, n, J, "float32")
    monkeypatch.setenv("CESX_POLL_JOIN", "1")
    monkeypatch.setenv("CESX_POLL_TIMEOUT_MS", "20")
    monkeypatch.setenv("CESX_TEST_DROP_CHOL_SIGNAL", "2")
    e1, U1, c1, sh1 = _aldi_chain(eng_mod, d, p, n, J, "float32")
    assert e1.poll_recoveries() == 1 and getattr(sh1, "redone_begins", 0) == 0
    assert np.array_equal(U1, U0) and np.array_equal(c1, c0)
    e2, U2, c2, sh2 = _aldi_chain(eng_mod, d, p, n, J, "float32", pipelined=True)
    assert e2.poll_recoveries() == 1 and sh2.redone_begins == 1
    assert np.array_equal(U2, U0) and np.array_equal(c2, c0)


@pytest.mark.parametrize("fast", ["0", "1"])
def test_recovery_refreshes_the_forward_map_of_the_pipelined_begin(eng_mod, monkeypatch, fast):
    """A pipelined driver evaluates G = forward(U_next) and enqueues the next step's moments BEFORE it reads the result
    of the step that writes U_next.  When that step's polled join runs out it wrote nothing, cesx_result re-runs it --
    and the G the driver holds was computed from an unwritten ensemble: the redo of the pipelined ``begin`` must
    re-evaluate the forward map INTO that tensor (ShardedUpdate.begin(forward=) / begin_lineal(out=)), or the next
    ``finish`` subtracts K G for a stale G.  With a real forward map (G = A U, the device hook of utils.lineal; with and
    without the linear-map shortcut) the chain must equal the event-joined one bit for bit."""
    import torch
    from ces_amd.dist import ShardedSampler
    from ces_amd.utils import lineal
    p, n, J, T = 128, 96, 8192, 5
    d = _synthetic(p, n, J, seed=83)
    monkeypatch.setenv("CESX_LINEAL_FAST", fast)

    def chain():
        eng = eng_mod.Engine(p, n, J, dtype="float32", seed=11)
        smp = ShardedSampler(eng, p, n, J)
        smp.T = T
        U = smp.run(d["y"], d["U0"], lineal(d["A"]), d["Gamma"], d["mu"], d["sigma"], d["ustar"], update="aldi", t_tol=1e30)
        torch.cuda.synchronize()
        return eng, U.cpu().numpy().copy(), {k: np.array(v) for k, v in smp.metrics.items()}, smp
    monkeypatch.setenv("CESX_POLL_JOIN", "0")
    _, U0, m0, _ = chain()
    monkeypatch.setenv("CESX_POLL_JOIN", "1")
    monkeypatch.setenv("CESX_POLL_TIMEOUT_MS", "20")
    monkeypatch.setenv("CESX_TEST_DROP_CHOL_SIGNAL", "2")
    e1, U1, m1, s1 = chain()
    assert e1.poll_recoveries() == 1 and s1.sh.redone_begins == 1
    assert np.array_equal(U1, U0)
    for k in m0:
        assert np.array_equal(m1[k], m0[k]), k


def test_polled_join_only_below_the_side_streams_priority(eng_mod, monkeypatch):
    """A waiter in front of what it waits for in one hardware queue never ends, and streams of one priority level may
    share a queue: the caller's stream polls the factorisation's word only when its priority is strictly lower than the
    side stream's.  On a high-priority caller stream, and with six engines alive, the chain is bit-identical to the
    event-joined one (CESX_POLL_JOIN=0) and no poll runs out."""
    import torch
    p, n, J = 128, 96, 8192
    d = _synthetic(p, n, J, seed=82)
    monkeypatch.setenv("CESX_POLL_JOIN", "0")
    e0, U0, c0, _ = _aldi_chain(eng_mod, d, p, n, J, "float32")
    monkeypatch.setenv("CESX_POLL_JOIN", "1")
    monkeypatch.setenv("CESX_POLL_TIMEOUT_MS", "200")
    hi = torch.cuda.Stream(priority=-1)
    keep = [eng_mod.Engine(p, n, J, dtype="float32", seed=k) for k in range(5)]      # five more engines alive (their side streams too)
    e1, U1, c1, _ = _aldi_chain(eng_mod, d, p, n, J, "float32", stream=hi)
    assert e1.poll_recoveries() == 0
    assert np.array_equal(U1, U0) and np.array_equal(c1, c0)
    e2, U2, c2, _ = _aldi_chain(eng_mod, d, p, n, J, "float32")                         # default stream, six engines alive
    assert e2.poll_recoveries() == 0
    assert np.array_equal(U2, U0) and np.array_equal(c2, c0)
    del keep


def test_profile_modes_select_the_sampled_kernel(eng_mod):
    """cesx_profile_enable(h, 3 / 4): only the update / only the moments launches of a step carry kernel-bound events
    (bench.py samples the dominant kernel alone inside its timed region); 1: both; 2: the gap's two events, which
    cesx_profile_read does not count."""
    p, n, J = 128, 96, 8192
    d = _synthetic(p, n, J, seed=5)
    eng = eng_mod.Engine(p, n, J, dtype="float32", seed=2)
    eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
    ref = None
    for mode, want in ((1, (2, 1)), (3, (0, 1)), (4, (2, 0)), (2, (0, 0)), (False, (0, 0))):
        eng.profile_enable(mode)
        out = eng.step(eng_mod.step_params(update="aldi", step_index=3), U, G, xi=None, recenter=True)
        res = eng.result()
        eng.profile_enable(False)
        gap = eng.profile_gap() if mode in (1, 2) else None
        k1, k3 = eng.profile_read(0), eng.profile_read(1)
        assert (k1[1], k3[1]) == want, (mode, k1, k3)
        assert (k1[0] > 0) == (want[0] > 0) and (k3[0] > 0) == (want[1] > 0)
        if gap is not None:
            assert 0 < gap < 5.0
        key = (out.cpu().numpy().copy(), res.hk)
        if ref is None:
            ref = key
        assert np.array_equal(key[0], ref[0]) and key[1] == ref[1]          # (time stamps do not change the numbers)


@pytest.mark.parametrize("update", ["aldi", "eks"])
def test_pde_model_run_drop_in(eng_mod, update):
    """A ``type == 'pde'`` forward model (Lorenz '63 with carried state W0)
    through the drop-in ``sampling.run``: same seeds as the reference run of
    oracle/make_golden_models.py -> same ensembles, carried states and metrics."""
    from ces_amd.calibrate import sampling
    from ces_amd.utils import lorenz63
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pde_run.npz"))
    model = lorenz63(l_window=int(g["l_window"]), freq=int(g["freq"]))
    p, J = g["U0"].shape
    eks = sampling(p=p, n_obs=model.n_obs, J=J)
    eks.ustar, eks.mu, eks.sigma, eks.T = g["ustar"], g["mu"], g["sigma"], int(g["T"])
    eks.parallel, eks.mute_bar = False, True
    np.random.seed(int(g["seed"]))
    eks.run(g["y"], np.copy(g["U0"]), model, g["Gamma"], np.linalg.cholesky(g["Gamma"]), wt=g["wt"], t=g["t"],
            update=update, t_tol=1e9)
    assert eks.Uall.shape == g[update + "_Uall"].shape and eks.Gall.shape == g[update + "_Gall"].shape
    # the ODE solves amplify the update's last-bit differences a little: 1e-6 on the ensembles
    assert rel_err(eks.Uall, g[update + "_Uall"]) < 1e-6
    assert rel_err(eks.Gall, g[update + "_Gall"]) < 1e-5
    assert rel_err(eks.W0, g[update + "_W0"]) < 1e-5 and rel_err(eks.Gstar, g[update + "_Gstar"]) < 1e-5
    for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t"):
        assert np.allclose(eks.metrics[k], g[update + "_metric_" + k], rtol=1e-5), k


def test_c_abi_argument_checks(eng_mod):
    """Error behaviour of the C ABI itself (include/cesx.h): bad shapes, aliasing, call order."""
    import ctypes as C
    with pytest.raises(eng_mod.CesxError):
        eng_mod.Engine(0, 3, 16)                                   # p < 1
    with pytest.raises(eng_mod.CesxError):
        eng_mod.Engine(20000, 3, 16)                               # p > 16384
    eng = eng_mod.Engine(4, 3, 64, dtype="float64")
    res = eng_mod.StepResult()
    assert eng.lib.cesx_result(eng._h, C.byref(res)) == eng_mod.ESTATE      # no step enqueued yet
    d = _synthetic(4, 3, 64, seed=1)
    eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    U, G = eng.to_device(d["U0"]), eng.to_device(d["G"])
    prm = eng_mod.step_params(update="aldi")
    with pytest.raises(ValueError, match="alias"):
        eng.step(prm, U, G, xi=None, out=U)                        # U_next must not alias U (ces/calibrate.py:357)
    with pytest.raises(ValueError):
        eng.step(prm, U[:, :32].contiguous(), G, xi=None)          # wrong shard width
    bad = eng_mod.step_params(update="aldi")
    bad.update = 7
    with pytest.raises(ValueError, match="update"):
        eng.step(bad, U, G, xi=None)
    assert eng.lib.cesx_abi_version() == eng_mod.ABI_VERSION
    out = eng.step(prm, U, G, xi=None)                             # the handle is still usable afterwards
    assert np.isfinite(eng.result().hk) and out.shape == (4, 64)


def test_rank_deficient_ensemble_fp32_request_runs_in_fp64(eng_mod):
    """J - 1 < p: the covariance is PD only through the 1e-8 jitter (ces/calibrate.py:476); the drop-in
    class then uses the fp64 engine even when 'float32' is requested, and matches the oracle."""
    from ces_amd.calibrate import sampling
    from oracle import ces_numpy as oc
    p, n, J = 40, 6, 20
    d = _synthetic(p, n, J, seed=41)
    eks = sampling(p=p, n_obs=n, J=J)
    eks.mu, eks.sigma, eks.ustar = d["mu"], d["sigma"], d["ustar"]
    eks.engine_dtype = "float32"
    eks.Uall = [d["U0"]]
    Uk = eks.eks_update_aldi(d["y"], d["U0"], d["G"], d["Gamma"], 0, xi=d["xi"])
    st = oc.OracleState(p, n, J, d["mu"], d["sigma"], d["ustar"])
    want = oc.literal_step(st, d["y"], d["U0"], d["G"], d["Gamma"], d["xi"], update="aldi")
    assert rel_err(Uk, want) < 1e-7


def test_in_place_modified_ensemble_gets_a_fresh_centring_pass(eng_mod):
    """The drop-in class skips the centring pass for an ensemble the engine itself produced (K2 predicted its
    mean).  A caller that edits the returned array IN PLACE -- here: moves every particle by 300 standard
    deviations -- must not be served with the stale shift: with it the fp32 second moments lose all their
    digits (cancellation); the 4096-sample check of sampling._device_update sees the edit and recentres."""
    from ces_amd.calibrate import sampling
    from oracle import ces_numpy as oc
    rng = np.random.default_rng(5)
    p, n, J = 16, 12, 4096
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    Gamma, sigma, mu = 0.01 * np.eye(n), 100.0 * np.eye(p), np.zeros((p, 1))
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    eks = sampling(p=p, n_obs=n, J=J)
    eks.mu, eks.sigma, eks.ustar = mu, sigma, ustar
    eks.engine_dtype = "float32"
    eks.Uall = [None]                                   # "first step" for the reference's len(self.Uall) == 1 test
    U0 = rng.standard_normal((p, J))
    xi0, xi1 = rng.standard_normal((p, J)), rng.standard_normal((p, J))
    U1 = eks.eks_update_aldi(y, U0, A @ U0, Gamma, 0, xi=xi0)
    assert isinstance(U1, np.ndarray)
    U1 += 300.0                                         # in place: the object identity survives, the contents do not
    eks.Uall.append(None)
    U2 = eks.eks_update_aldi(y, U1, A @ U1, Gamma, 1, xi=xi1)
    st = oc.OracleState(p, n, J, mu, sigma, ustar)
    st.trace_len = 2
    st.metrics["t"].append(eks.metrics["t"][0])
    cast = lambda a: a.astype(np.float32).astype(np.float64)
    ref = oc.factored_step(st, y, cast(U1), cast(A @ U1), Gamma, cast(xi1), update="aldi")
    assert rel_err(U2, ref) < TOL32
    assert eks.metrics["bias"][-1] == pytest.approx(st.metrics["bias"][-1], rel=TOL32)


@pytest.mark.parametrize("shape,dtype,tol", [((20, 12, 1024, 6), "float64", 1e-10), ((64, 50, 4096, 6), "float64", 1e-10),
                                             ((256, 256, 4096, 5), "float32", 2e-4), ((96, 160, 2048, 5), "float32", 2e-4)])
def test_lineal_fast_path_matches_the_full_gram(eng_mod, monkeypatch, shape, dtype, tol):
    """Chained device-resident loop with a linear forward map on the device: the G-dependent moments taken from the
    U-only head and the installed map (cesx_moments_rest_lineal, no second Gram launch) against the same loop with
    the full Gram over G (CESX_LINEAL_FAST=0), with an offset b and shapes that are not multiples of the MFMA tile;
    the fp64 run is also held against the pinned oracle."""
    from ces_amd.dist import ShardedSampler
    from ces_amd.utils import lineal
    from oracle import ces_numpy as oc
    p, n, J, T = shape
    rng = np.random.default_rng(21)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    b = 0.3 * rng.standard_normal(n)
    ustar = rng.standard_normal((p, 1))
    Gamma, sigma, mu = 0.01 * np.eye(n), 100.0 * np.eye(p), np.zeros((p, 1))
    y = (A @ ustar).ravel() + b + 0.1 * rng.standard_normal(n)
    U0 = ustar + rng.standard_normal((p, J))
    xis = rng.standard_normal((T, p, J))
    outs = []
    for fast in ("1", "0"):
        monkeypatch.setenv("CESX_LINEAL_FAST", fast)
        eng = eng_mod.Engine(p, n, J, dtype=dtype, seed=4)
        smp = ShardedSampler(eng, p, n, J)
        smp.T = T
        model = lineal(A, b)
        assert smp.sh.lineal_fast_ok(model) == (fast == "1")
        U = smp.run(y, U0, model, Gamma, mu, sigma, ustar, xis=xis, t_tol=1e9)
        outs.append((U.cpu().numpy().astype(np.float64), {k: list(v) for k, v in smp.metrics.items()}))
    assert rel_err(outs[0][0], outs[1][0]) < tol
    for k in outs[0][1]:
        assert np.allclose(outs[0][1][k], outs[1][1][k], rtol=max(tol, 1e-9) * 10), k
    if dtype == "float64":
        st = oc.OracleState(p, n, J, mu, sigma, ustar, T=T)
        Uall, _ = oc.run_chain(st, y, U0, lambda U: A @ U + b[:, None], Gamma, xis, update="aldi", step=oc.factored_step, t_tol=1e9)
        assert rel_err(outs[0][0], Uall[-1]) < TOL64
        assert np.allclose(outs[0][1]["t"], st.metrics["t"], rtol=1e-8)


@pytest.mark.parametrize("update,dtype,trace", [("aldi", "float32", False), ("aldi", "float64", True), ("eks", "float32", True),
                                                ("aldi_constant", "float32", False)])
def test_pipelined_host_loop_equals_the_plain_host_loop(eng_mod, update, dtype, trace):
    """``sampling.run`` with a HOST forward map on a large ensemble: the loop pipelined over column blocks
    (G_ens block by block, G up while the next block is evaluated, the ensemble never uploaded again) gives the
    same bits as the reference's flow (one G_ens call, float64 arrays into and out of every update)."""
    from ces_amd.calibrate import sampling
    p, n, J, T = 256, 200, 16388, 4                      # p J >= 2^22 (the pipelined loop's threshold), ragged blocks
    rng = np.random.default_rng(8)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    Gamma, sigma, mu = 0.01 * np.eye(n), 100.0 * np.eye(p), np.zeros((p, 1))
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    U0 = ustar + rng.standard_normal((p, J))

    class host_model:
        type, model_name, n_obs = "map", "lineal", n

        def __call__(self, theta):
            return A @ theta + 0.05 * np.tanh(theta[:n])          # mildly nonlinear, per particle
    outs = []
    for pipe in (True, False):
        eks = sampling(p=p, n_obs=n, J=J)
        eks.mu, eks.sigma, eks.ustar = mu, sigma, ustar
        eks.engine_dtype, eks.noise, eks.T, eks.seed = dtype, "device", T, 5
        eks.host_pipeline = pipe
        calls = []
        g_ens = eks.G_ens

        def counted(theta, m, g_ens=g_ens, calls=calls):
            calls.append(theta.shape[1])
            return A @ theta + 0.05 * np.tanh(theta[:n])        # (= m(theta[:, j]) for every particle j)
        eks.G_ens = counted
        eks.run(y, U0.copy(), host_model(), Gamma, None, trace=trace, update=update, t_tol=1e30)
        assert (len(calls) == 8 * T + 1) == pipe and sum(calls) == (T + 1) * J
        outs.append((eks.Ustar.copy(), eks.Gstar.copy(), {k: list(v) for k, v in eks.metrics.items()},
                     np.array(eks.Uall) if trace else None, np.array(eks.Gall) if trace else None))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]
    if trace:
        assert np.array_equal(outs[0][3], outs[1][3]) and np.array_equal(outs[0][4], outs[1][4])
