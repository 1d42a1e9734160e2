"""The CPU oracle (oracle/ces_numpy.py) against the reference's own outputs.

The fixtures under tests/golden/ were produced by the real reference
(ces/calibrate.py:270-529) through oracle/make_golden.py; these tests pin both
oracle forms to them so that the GPU parity tests can trust the oracle.
"""
import numpy as np
import pytest

from oracle import ces_numpy as oc
from conftest import step_case, rel_err

TOL_LITERAL = 1e-12
TOL_FACTORED = 1e-9     # different but equivalent algebra, fp64


def _state(case, c):
    st = oc.OracleState(case["p"], case["n_obs"], case["J"], c["mu"], c["sigma"], c["ustar"])
    st.trace_len = case["trace_len"]
    st.metrics["t"] = list(case["t_prev"])
    return st


def _check(st, Uk, case, c, tol):
    assert rel_err(Uk, c["Uk"]) < tol
    assert abs(st.metrics["t"][-1] - float(c["t_new"])) <= tol * max(1.0, abs(float(c["t_new"])))
    got = np.array([st.metrics[k][-1] for k in ("self-bias", "self-bias-data", "bias-data", "bias")])
    assert np.allclose(got, c["metrics"], rtol=max(tol, 1e-10), atol=0)
    if case["kwargs"].get("time_step") == "spectral":
        assert np.allclose(st.radspec, c["radspec"], rtol=1e-8)
    assert st.update_rule == case["update_rule"]


def test_manifest_covers_the_matrix(manifest):
    steps = manifest["steps"]
    assert len(steps) >= 100
    assert {s["update"] for s in steps} == {"eks", "aldi", "aldi_constant"}
    assert {s["time_step_case"] for s in steps} >= {"default", "spectral", "constant", "mix_spinup",
                                                     "mix_spinup_late", "mix_after"}
    assert any(s["dense_gamma"] and s["dense_sigma"] and s["nonlinear_G"] for s in steps)


def test_literal_matches_reference(manifest, golden_steps):
    for case in manifest["steps"]:
        c = step_case(golden_steps, case)
        st = _state(case, c)
        U0 = c["U0"].copy()
        Uk = oc.literal_step(st, c["y"], U0, c["G"], c["Gamma"], c["xi"],
                             update=case["update"], **case["kwargs"])
        assert np.array_equal(U0, c["U0"]) and Uk is not U0
        _check(st, Uk, case, c, TOL_LITERAL)


def test_factored_matches_reference(manifest, golden_steps):
    for case in manifest["steps"]:
        c = step_case(golden_steps, case)
        st = _state(case, c)
        Uk = oc.factored_step(st, c["y"], c["U0"], c["G"], c["Gamma"], c["xi"],
                              update=case["update"], **case["kwargs"])
        _check(st, Uk, case, c, TOL_FACTORED)


def test_factored_fp32_within_north_star_tolerance(manifest, golden_steps):
    # BASELINE.json north_star: 1e-3 relative in fp32
    worst = 0.0
    for case in manifest["steps"]:
        c = step_case(golden_steps, case)
        st = _state(case, c)
        Uk = oc.factored_step(st, c["y"], c["U0"], c["G"], c["Gamma"], c["xi"],
                              update=case["update"], dtype=np.float32, **case["kwargs"])
        assert Uk.dtype == np.float32
        worst = max(worst, rel_err(Uk, c["Uk"]))
    assert worst < 1e-3


@pytest.mark.parametrize("step", [oc.literal_step, oc.factored_step])
def test_error_paths(manifest, golden_errors, step):
    for e in manifest["errors"]:
        c = {k.split("_", 1)[1]: v for k, v in golden_errors.items() if k.startswith("e%d_" % e["id"])}
        st = oc.OracleState(e["p"], e["n_obs"], e["J"], c["mu"], c["sigma"], c["ustar"])
        xi = np.zeros((e["p"], e["J"]))
        expected = {"LinAlgError": np.linalg.LinAlgError, "AttributeError": AttributeError,
                    "UnboundLocalError": UnboundLocalError}[e["error"]]
        with pytest.raises(expected):
            step(st, c["y"], c["U0"], c["G"], c["Gamma"], xi, update=e["update"], **e["kwargs"])


@pytest.mark.parametrize("update", ["aldi", "eks", "aldi_constant"])
@pytest.mark.parametrize("step,tol", [(oc.literal_step, 1e-10), (oc.factored_step, 1e-7)])
def test_trajectory_c1(manifest, golden_traj, update, step, tol):
    """BASELINE.json configs[0]: J=100, p=2, n_obs=10, 30 iterations of
    sampling.run with utils.lineal, noise injected per step."""
    info = next(t for t in manifest["trajectories"] if t["update"] == update)
    g = golden_traj
    st = oc.OracleState(info["p"], info["n_obs"], info["J"], g["mu"], g["sigma"], g["ustar"], T=info["T"])
    fwd = lambda U: oc.lineal_forward(g["A"], U)
    Uall, Gall = oc.run_chain(st, g["y"], g[update + "_U0"], fwd, g["Gamma"], g[update + "_xis"],
                              update=update, step=step, t_tol=1e9)
    assert Uall.shape == g[update + "_Uall"].shape == (info["steps"] + 1, info["p"], info["J"])
    assert rel_err(Uall, g[update + "_Uall"]) < tol
    assert rel_err(Gall, g[update + "_Gall"]) < tol
    for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t"):
        assert np.allclose(st.metrics[k], g[update + "_m_" + k], rtol=max(tol, 1e-9))


def test_c1_posterior_sanity_band(golden_traj):
    """Known answer from examples/notebooks/linear.ipynb:692-697 (analytic
    posterior mean [-1.0367, 2.0870]); sampling error only allows a band."""
    U = golden_traj["aldi_Uall"][-1]
    assert np.allclose(U.mean(axis=1), [-1.03673079, 2.08697021], atol=0.15)


def test_philox_oracle_matches_random123_known_answers():
    """oracle/philox.py restates Philox4x32-10; these are the three known-answer vectors Random123 ships for it
    (kat_vectors: counter x 4, key x 2 -> output x 4).  The device generator is compared against this oracle
    in the -m gpu tests, so the chain device == oracle == published algorithm is closed here."""
    from oracle import philox
    kat = [
        ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
         (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want, (ctr, key)
    # vectorised call = element-wise calls
    ctrs = np.array([k[0] for k in kat], dtype=np.uint32).T
    got = philox.philox4x32_10(ctrs[0][:1], ctrs[1][:1], ctrs[2][:1], ctrs[3][:1], kat[0][1][0], kat[0][1][1])
    assert int(got[0][0]) == kat[0][2][0]
