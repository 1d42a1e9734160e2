"""Darcy-flow forward map on the host, in numpy/scipy.

The reference drives two MATLAB files through ``matlab.engine``
(``ces/darcy.py:9-138``); MATLAB is proprietary and absent here, so this module
restates their arithmetic:

* ``gaussrnd_coarse`` (utilities/mfiles/gaussrnd_coarse.m:6-23): KL
  coefficients ``N * tau^(alpha-1) (pi^2 |k|^2 + tau^2)^(-alpha/2) * xi`` with the
  constant mode removed, synthesised by MATLAB's ``idct2`` = orthonormal inverse
  2-D DCT-II (``scipy.fft.idctn(..., norm='ortho')``).
* ``solve_gwf`` (utilities/mfiles/solve_gwf.m:4-38): ``-div(exp(a) grad p) = 1`` on
  the unit square, ``p = 0`` on the boundary: cell-centred log-permeability is
  spline-interpolated to the K x K node grid (MATLAB ``interp2(...,'spline')`` =
  tensor-product not-a-knot cubic splines, which extrapolate to the boundary
  nodes), 5-point finite differences with arithmetic-mean face coefficients on
  the (K-2)^2 interior nodes, sparse solve, and spline interpolation of the
  pressure back to the cell centres.

Same classes and call conventions as the reference (``model``, ``model_trunc``,
``set_initial``, ``set_rank``, ``eval_rf``, ``solve_pde``, ``obs_index``);
``start`` / ``stop`` / ``set_rnd_seed`` are kept as no-ops so that
examples/scripts/darcy-flow.py:6-14 runs unchanged.  The forward map stays on
the host (BASELINE.json north_star); only the ensemble update runs on the GPU.

PARITY UNPINNED: the MATLAB reference cannot run here, so there is no golden
output; tests validate the restatement through the PDE itself (discrete
residual, manufactured solution, symmetry) rather than against the reference.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.fft import idctn
from scipy.interpolate import CubicSpline


def gaussrnd_coarse(xi, alpha, tau, N):
    """utilities/mfiles/gaussrnd_coarse.m:6-23.  ``xi``: (N, N) KL coefficients."""
    N = int(N)
    k = np.arange(N)
    K1, K2 = np.meshgrid(k, k)
    coef = tau ** (alpha - 1) * (np.pi ** 2 * (K1 ** 2 + K2 ** 2) + tau ** 2) ** (-alpha / 2)
    L = N * coef * np.asarray(xi, dtype=np.float64).reshape(N, N)
    L[0, 0] = 0.0
    return idctn(L, norm="ortho")


def _interp2_spline(x_src, z, x_dst):
    """MATLAB interp2(X, Y, Z, Xq, Yq, 'spline') on tensor grids with the same 1-D
    abscissae in both directions: not-a-knot cubic splines along each axis."""
    z = CubicSpline(x_src, z, axis=0, bc_type="not-a-knot", extrapolate=True)(x_dst)
    return CubicSpline(x_src, z, axis=1, bc_type="not-a-knot", extrapolate=True)(x_dst)


def assemble_gwf(a_nodes):
    """5-point operator of solve_gwf.m:19-35 for nodal coefficients ``a_nodes`` (K, K):
    returns the sparse matrix acting on the interior unknowns ordered column by column
    (MATLAB ``F(:)`` order), already scaled by (K-1)^2."""
    K = a_nodes.shape[0]
    m = K - 2
    c = a_nodes
    idx = lambda i, j: (j - 1) * m + (i - 1)          # interior node (i, j), 1-based interior indices
    rows, cols, vals = [], [], []
    for j in range(1, K - 1):
        for i in range(1, K - 1):
            diag = ((c[i - 1, j] + c[i, j]) / 2 + (c[i + 1, j] + c[i, j]) / 2
                    + (c[i, j - 1] + c[i, j]) / 2 + (c[i, j + 1] + c[i, j]) / 2)
            rows.append(idx(i, j)); cols.append(idx(i, j)); vals.append(diag)
            if i > 1:
                rows.append(idx(i, j)); cols.append(idx(i - 1, j)); vals.append(-(c[i - 1, j] + c[i, j]) / 2)
            if i < K - 2:
                rows.append(idx(i, j)); cols.append(idx(i + 1, j)); vals.append(-(c[i + 1, j] + c[i, j]) / 2)
            if j > 1:
                rows.append(idx(i, j)); cols.append(idx(i, j - 1)); vals.append(-(c[i, j - 1] + c[i, j]) / 2)
            if j < K - 2:
                rows.append(idx(i, j)); cols.append(idx(i, j + 1)); vals.append(-(c[i, j + 1] + c[i, j]) / 2)
    return sp.csc_matrix((vals, (rows, cols)), shape=(m * m, m * m)) * (K - 1) ** 2


def solve_gwf(coef):
    """utilities/mfiles/solve_gwf.m:4-38.  ``coef``: (K, K) log-permeability at cell
    centres; returns the pressure at the cell centres, (K, K)."""
    coef = np.asarray(coef, dtype=np.float64)
    K = coef.shape[0]
    centres = np.arange(1, 2 * K, 2) / (2.0 * K)       # 1/(2K) : 1/K : (2K-1)/(2K)
    nodes = np.linspace(0.0, 1.0, K)
    a = _interp2_spline(centres, np.exp(coef), nodes)
    A = assemble_gwf(a)
    rhs = np.ones((K - 2) * (K - 2))                    # spline of the constant 1 is 1
    x = spla.spsolve(A, rhs)
    P = np.zeros((K, K))
    P[1:-1, 1:-1] = x.reshape(K - 2, K - 2, order="F")  # x is in column-major (MATLAB F(:)) order
    # solve_gwf.m builds vec2mat(x, K-2) (= P^T), interpolates and transposes back; the grids
    # are the same in both directions, so that equals interpolating P itself
    return _interp2_spline(nodes, P, centres)


class model(object):
    """Full KL parametrisation, p = Nmesh^2 (ces/darcy.py:9-98)."""

    def __init__(self, alpha=2., tau=3., Nmesh=2. ** 4):
        self.alpha = alpha
        self.tau = tau
        self.Nmesh = Nmesh
        self.p = int(self.Nmesh * self.Nmesh)
        self.model_name = 'darcy-flow'
        self.type = 'map'

    def __call__(self, xi, full_solution=False):
        theta = self.eval_rf(xi)
        U = self.solve_pde(theta)
        if full_solution:
            return np.asarray(U).flatten()
        return np.asarray(U).flatten()[self.obs_index]

    # the MATLAB engine life cycle of ces/darcy.py:40-66 has no counterpart here
    def start(self, mpath=None):
        pass

    def stop(self):
        pass

    def set_rnd_seed(self, seed=1):
        pass

    def set_initial(self, seed=1):
        np.random.seed(seed)
        self.ustar = np.random.normal(0, 1, int(self.p))

    def set_rank(self):
        k = np.arange(int(self.Nmesh))
        K1, K2 = np.meshgrid(k, k)
        self.eigs = (self.tau ** (self.alpha - 1)) * (np.pi ** 2 * (K1 ** 2 + K2 ** 2) + self.tau ** 2) ** (-self.alpha / 2)
        self.eigs[0, 0] = 1e-10
        self.rank = (-self.eigs).flatten().argsort()

    def eval_rf(self, xi):
        return gaussrnd_coarse(np.asarray(xi, dtype=np.float64).reshape(int(self.Nmesh), -1),
                               self.alpha, self.tau, self.Nmesh)

    def solve_pde(self, theta):
        return solve_gwf(theta)


class model_trunc(model):
    """Rank-ordered truncated KL expansion (ces/darcy.py:100-138)."""

    def __init__(self, alpha=2., tau=3., Nmesh=2. ** 4, p=10):
        super().__init__(alpha=alpha, tau=tau, Nmesh=Nmesh)
        super().set_rank()
        self.p = p

    def set_initial(self, seed=1):
        np.random.seed(seed)
        ustar = np.random.normal(0, 1, int(self.Nmesh * self.Nmesh))
        self.ustar = ustar[self.rank[:self.p]]

    def eval_rf(self, xi):
        full = np.zeros(int(self.Nmesh * self.Nmesh))
        full[self.rank[:self.p]] = np.copy(xi)
        return gaussrnd_coarse(full.reshape(int(self.Nmesh), -1), self.alpha, self.tau, self.Nmesh)
