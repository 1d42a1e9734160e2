"""Host Darcy forward map (ces_amd/darcy.py) restating utilities/mfiles/*.m.
PARITY UNPINNED (no MATLAB here): validated through the PDE itself."""
import numpy as np
import pytest

from ces_amd import darcy


def test_gaussrnd_is_orthonormal_dct_synthesis():
    rng = np.random.default_rng(0)
    N, alpha, tau = 16, 2.0, 3.0
    xi = rng.standard_normal((N, N))
    U = darcy.gaussrnd_coarse(xi, alpha, tau, N)
    k = np.arange(N)
    K1, K2 = np.meshgrid(k, k)
    L = N * tau ** (alpha - 1) * (np.pi ** 2 * (K1 ** 2 + K2 ** 2) + tau ** 2) ** (-alpha / 2) * xi
    L[0, 0] = 0
    assert abs(U.mean()) < 1e-12                                   # constant mode removed (gaussrnd_coarse.m:20)
    assert np.isclose((U ** 2).sum(), (L ** 2).sum())              # idct2 is orthonormal
    # explicit DCT-II synthesis of one mode
    e = np.zeros((N, N)); e[2, 3] = 1.0
    Ue = darcy.gaussrnd_coarse(e / (N * tau ** (alpha - 1) * (np.pi ** 2 * 13 + tau ** 2) ** (-alpha / 2)), alpha, tau, N)
    x = (2 * np.arange(N) + 1) * np.pi / (2 * N)
    want = (2.0 / N) * np.outer(np.cos(2 * x), np.cos(3 * x))
    assert np.allclose(Ue, want, atol=1e-12)


def test_operator_is_symmetric_positive_definite_and_solved_exactly():
    rng = np.random.default_rng(1)
    K = 16
    a = np.exp(0.5 * rng.standard_normal((K, K)))
    A = darcy.assemble_gwf(a)
    assert abs(A - A.T).max() < 1e-9
    import scipy.sparse.linalg as spla
    assert spla.eigsh(A, k=1, which="SA", return_eigenvectors=False)[0] > 0


@pytest.mark.parametrize("K,tol", [(16, 0.05), (48, 0.01)])
def test_constant_permeability_matches_poisson_series(K, tol):
    """-Laplace p = 1, p = 0 on the boundary: p(1/2, 1/2) = 0.0736713..."""
    P = darcy.solve_gwf(np.zeros((K, K)))
    centres = np.arange(1, 2 * K, 2) / (2.0 * K)
    # series solution at the cell centres
    x, y = np.meshgrid(centres, centres, indexing="ij")
    ref = np.zeros_like(x)
    for m in range(1, 60, 2):
        for n in range(1, 60, 2):
            ref += 16.0 / (np.pi ** 4 * m * n * (m * m + n * n)) * np.sin(m * np.pi * x) * np.sin(n * np.pi * y)
    assert np.abs(P - ref).max() < tol * ref.max()
    assert np.allclose(P, P.T, atol=1e-12) and np.allclose(P, P[::-1, :], atol=1e-10)


def test_transposed_field_gives_transposed_pressure():
    rng = np.random.default_rng(2)
    a = 0.3 * rng.standard_normal((16, 16))
    assert np.allclose(darcy.solve_gwf(a.T), darcy.solve_gwf(a).T, atol=1e-10)


def test_model_classes_follow_reference_conventions():
    """ces/darcy.py:9-138 as used by examples/scripts/darcy-flow.py:6-41."""
    m = darcy.model()
    assert (m.p, m.type, m.model_name) == (256, "map", "darcy-flow")
    m.start(mpath="./mfiles"); m.set_rnd_seed(); m.set_initial()
    assert m.ustar.shape == (256,)
    m.n_obs = 50
    full = m(m.ustar, full_solution=True)
    assert full.shape == (256,) and np.all(np.isfinite(full)) and full.min() > -1e-3
    np.random.seed(1)
    m.obs_index = np.random.choice(m.p, m.n_obs, replace=False, p=full / full.sum())
    assert m(m.ustar).shape == (50,)
    t = darcy.model_trunc(p=64)
    t.set_initial()
    assert t.ustar.shape == (64,) and t.rank.shape == (256,)
    assert t.rank[-1] == 0 and set(t.rank[:2]) == {1, 16}        # constant mode ranked last (ces/darcy.py:80)
    t.obs_index = m.obs_index
    g = t(t.ustar)
    assert g.shape == (50,)
    # the truncated map with all modes kept equals the full map
    tt = darcy.model_trunc(p=256); tt.obs_index = m.obs_index
    xi = np.random.default_rng(3).standard_normal(256)
    assert np.allclose(tt(xi[tt.rank]), m(xi))
