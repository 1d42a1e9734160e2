"""Host forward models with carried state, the Lorenz family.

``sampling.run`` drives two kinds of forward maps (ces/calibrate.py:342-353):
``type == 'map'`` (one call per particle, ces/utils.py:5-122) and ``type ==
'pde'`` (an ODE solved per particle from a carried state, then summarised into
observables: ces/calibrate.py:132-168 and ces/utils.py:124-455).  These are host
code in the reference and stay host code here (BASELINE.json north_star); only
the ensemble update runs on the GPU.  The classes keep the reference's names,
attributes, call conventions and default arguments so that its notebooks
(examples/notebooks/{linear,lorenz63}.ipynb) run against ``ces_amd.calibrate``
unchanged; the right-hand sides are written in vectorised numpy rather than
per-variable Python loops.  ``lineal_log``, ``elliptic`` and ``banana`` of
ces/utils.py are outside the hot-path scope and are
not shipped.

Parity: every class is checked against outputs of the reference's own
``ces/utils.py`` (it imports unmodified in the build container) stored in
tests/golden/models.npz by oracle/make_golden_models.py.
"""
import numpy as np

from .utils import lineal


class lorenz63(object):
    """Lorenz '63 with (r, b) as parameters, sigma = 10 fixed (ces/utils.py:124-194).

    ``solve`` integrates with ``scipy.integrate.odeint`` over the time vector the driver
    passes (ces/calibrate.py:145); ``statistics`` returns the 9 first and second moments
    over the LAST window of ``l_window * freq`` samples (the sample at t[0] is dropped,
    ces/utils.py:191-193)."""

    def __init__(self, l_window=10, freq=100):
        self.n_state = 3
        self.n_obs = 9
        self.l_window = l_window
        self.freq = freq
        self.solve_init = False
        self.model_name = "lorenz63"
        self.type = "pde"

    def __repr__(self):
        return self.model_name

    def __str__(self):
        return self.model_name + str(self.n_state)

    def model(self, w, t, sigma=10., r=28., b=8. / 3):
        x, y, z = w
        return [sigma * (y - x), r * x - y - x * z, x * y - b * z]

    def __call__(self, w, t, r=28., b=8. / 3):
        return self.model(w, t, 10., r, b)

    def solve(self, w0, t, args=()):
        from scipy import integrate
        return integrate.odeint(self, w0, t, args=args)

    def statistics(self, ws):
        x, y, z = ws[:, 0], ws[:, 1], ws[:, 2]
        feats = np.stack([x, y, z, x * x, y * y, z * z, x * y, x * z, y * z])
        win = int(self.l_window * self.freq)
        # adjacent windows over samples 1.. ; the last one is the observable
        return feats[:, 1:].reshape(self.n_obs, -1, win).mean(axis=2)[:, -1]


class lorenz63_log(lorenz63):
    """Same system in (log r, log b) (ces/utils.py:196-227)."""

    def __init__(self, l_window=10, freq=100):
        super().__init__(l_window=l_window, freq=freq)
        self.model_name = "lorenz63_log"

    def model(self, w, t, sigma=10., log_r=np.log(28.), log_b=np.log(8. / 3)):
        return super().model(w, t, sigma, np.exp(log_r), np.exp(log_b))

    def __call__(self, w, t, log_r=np.log(28.), log_b=np.log(8. / 3)):
        return self.model(w, t, 10., log_r, log_b)

    def grad_logjacobian(self, params):
        return -np.exp(-params)

    def logjacobian(self, params):
        return -params.sum(axis=0)


def _l96_rhs(X, n_slow, n_fast, h, F, c, b):
    """Two-scale Lorenz '96 tendencies (ces/utils.py:279-297), vectorised with cyclic shifts:
    dX_k = -X_{k-1}(X_{k-2} - X_{k+1}) - X_k + F - h c mean_l(Y_{l,k}),
    dY_j = -c b Y_{j+1}(Y_{j+2} - Y_{j-1}) - c Y_j + (h c / n_fast) X_{j // n_fast}."""
    Y = X[n_slow:]
    X = X[:n_slow]
    dX = (-np.roll(X, 1) * (np.roll(X, 2) - np.roll(X, -1)) - X + F
          - (h * c) * Y.reshape(n_slow, n_fast).mean(axis=1))
    dY = (-c * b * np.roll(Y, -1) * (np.roll(Y, -2) - np.roll(Y, 1)) - c * Y
          + ((h * c) / n_fast) * np.repeat(X, n_fast))
    return np.hstack((dX, dY))


class lorenz96(object):
    """Two-scale Lorenz '96, parameters (h, F, log c, b) (ces/utils.py:229-343)."""

    n_params = 4

    def __init__(self, n_slow=36, n_fast=10, l_window=10, freq=10, spinup=10):
        self.n_slow = n_slow
        self.n_fast = n_fast
        self.n_state = self.n_slow * (self.n_fast + 1)
        self.l_window = l_window
        self.freq = freq
        self.spinup = spinup
        self.solve_init = False
        self.model_name = "lorenz96"
        self.type = "pde"

    def __repr__(self):
        tail = "" if self.n_params == 4 else "," + str(self.n_params)
        return self.model_name + "," + str(self.n_slow) + "," + str(self.n_fast) + tail

    def __str__(self):
        print("Model: ..................... Lorenz 96")
        print("Number of slow variables ... %s" % (self.n_slow))
        print("Number of fast variables ... %s" % (self.n_fast))
        print("Number of parameters........ %s" % (self.n_params))
        print("Solver initialized ......... %s" % (self.solve_init))
        return str()

    def model(self, X, t, h=1., F=10., log_c=np.log(10.), b=10.):
        return _l96_rhs(np.asarray(X), self.n_slow, self.n_fast, h, F, np.exp(log_c), b)

    def __call__(self, t, w, h=1., F=10., log_c=np.log(10.), b=10.):
        return self.model(w, t, h, F, log_c, b)

    def generate_initial(self):
        """Slow variables ~ U(-5, 10); every fast variable starts at its slow variable."""
        x = np.random.rand(self.n_slow) * 15 - 5
        return np.concatenate([x, np.repeat(x, self.n_fast)])

    def set_solver(self, method="RK45", T=20, dt=0.1):
        self.method, self.dt, self.T = method, dt, T
        self.solve_init = True

    def solve(self, w0, t, args=()):
        from scipy import integrate
        if not self.solve_init:
            raise TypeError("lorenz96.solve: call set_solver first")   # the reference fails in np.empty() here
        res = integrate.solve_ivp(fun=lambda tt, y: self(tt, y, *args), t_span=[0, self.T], y0=w0,
                                  t_eval=t, method=self.method, max_step=self.dt)
        return res.y.T

    def _phi(self, ws):
        """5 blocks of n_slow window means: X, X^2, mean_l Y, mean_l Y^2, X mean_l Y
        (ces/utils.py:331-341); columns = adjacent windows after the spin-up."""
        ws = ws.T
        win = self.l_window * self.freq
        data = ws[:, (self.spinup * self.freq + 1):].reshape(self.n_state, -1, win)
        X = data[:self.n_slow]
        Y = data[self.n_slow:].reshape(self.n_slow, self.n_fast, -1, win)
        Ybar = Y.mean(axis=1)
        return np.vstack([X.mean(axis=2), (X ** 2).mean(axis=2), Ybar.mean(axis=2),
                          (Y ** 2).mean(axis=1).mean(axis=2), (X * Ybar).mean(axis=2)])

    def statistics(self, ws):
        return self._phi(ws)[:, -1]

    def grad_logjacobian(self, params):
        # as the reference computes it (ces/utils.py:343-347): the exponent is taken of the zero it
        # has just written, so the third entry is -1
        out = np.zeros_like(params)
        out[2] = -np.exp(-out[2])
        return out


class lorenz96_hom(lorenz96):
    """Statistics averaged over the slow index (homogeneous), ces/utils.py:349-367."""

    def __init__(self):
        super().__init__()
        self.hom = True

    def statistics(self, ws):
        phi = self._phi(ws)[:, -1].reshape(5, -1)
        return phi.mean(axis=1) if self.hom else phi[:, 7]


class lorenz96Fc(lorenz96):
    """(F, log c) free, h = 1, b = 10 (ces/utils.py:369-389)."""
    n_params = 2

    def __init__(self):
        super().__init__()

    def __call__(self, t, w, F=10., log_c=np.log(10.)):
        return self.model(w, t, 1., F, log_c, 10.)


class lorenz96Fb(lorenz96):
    """(F, b) free (ces/utils.py:391-408)."""
    n_params = 2

    def __call__(self, t, w, F=10., b=10.):
        return self.model(w, t, 1., F, np.log(10), b)


class lorenz96hFb(lorenz96):
    """(h, F, b) free (ces/utils.py:410-428)."""
    n_params = 3

    def __call__(self, t, w, h=1., F=10., b=10.):
        return self.model(w, t, h, F, np.log(10), b)


class lorenz96hcb(lorenz96):
    """(h, log c, b) free (ces/utils.py:430-448)."""
    n_params = 3

    def __call__(self, t, w, h=1., log_c=np.log(10.), b=10.):
        return self.model(w, t, h, 10., log_c, b)


def lorenz96_dim(t, X, h=1., F=10., c=2 ** 7., b=1.):
    """Dimensional two-scale Lorenz '96 with fixed coupling 0.8 (ces/utils.py:450-466)."""
    n_slow, n_fast = 36, 10
    Y = X[n_slow:]
    X = X[:n_slow]
    dX = -np.roll(X, 1) * (np.roll(X, 2) - np.roll(X, -1)) - X + F - 0.8 * Y.reshape(n_slow, n_fast).mean(axis=1)
    dY = -c * np.roll(Y, -1) * (np.roll(Y, -2) - np.roll(Y, 1)) - c * Y + c * np.repeat(X, n_fast)
    return np.hstack((dX, dY))
