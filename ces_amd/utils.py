"""Forward maps of the hot-path configs (ces/utils.py:5-31).

``lineal`` is the forward map of the path BASELINE.json names (configs 1-3) and the
one with a device hook; the other forward models of ces/utils.py (host code, SURVEY.md
8f rank 4) live in ces_amd/models.py and are re-exported here, so that
``import ces_amd.utils as utils`` offers what ``import ces.utils as utils`` does.
"""
import numpy as np


class lineal(object):
    """Linear forward map ``A theta + b`` (ces/utils.py:5-31): same attributes
    (``A, b, n_obs, flag_noise, noise_sigma, model_name, type``) and call
    convention, one parameter vector per call as enka.G_ens uses it
    (ces/calibrate.py:123-130)."""

    engine_lineal = True       # build-only: forward_device evaluates G = A U + b with the map installed in the engine

    def __init__(self, A, b=0, flag_noise=False):
        self.A = A
        self.b = b
        self.n_obs = A.shape[0]
        self.flag_noise = flag_noise
        self.noise_sigma = np.sqrt(0.1)
        self.model_name = "lineal"
        self.type = "map"

    def __repr__(self):
        return self.model_name

    def __call__(self, theta):
        out = np.matmul(self.A, theta) + self.b
        if self.flag_noise:
            out = out + self.noise_sigma * np.random.normal()
        return out

    def _fingerprint(self):
        out = []
        for a in (self.A, self.b):
            a = np.asarray(a)
            flat = a.reshape(-1)
            out.append((a.shape, a.dtype.str, flat[::max(1, flat.size // 64)].tobytes(), float(flat.sum()) if flat.size else 0.0))
        return tuple(out)

    def invalidate_device(self):
        """Forget the map installed in an engine (build-only): the next ``forward_device`` uploads A and b again."""
        self._dev_A = self._dev_b = self._dev_fp = None
        self._dev_token = 0

    # build-only hook: evaluate the whole shard on device
    def forward_device(self, engine, U_dev, out=None):
        if self.flag_noise:
            raise ValueError("the device hook evaluates the noise-free map only")
        # The map is installed in the engine once (cesx_forward_set_lineal: the engine keeps A and b in the layout of
        # its LDS-DMA update kernels) and applied every iteration (a per-call upload of a pageable host array cost
        # 9 ms per step at p = n_obs = 256, 20x the ensemble update itself).  An engine without the two-step entry
        # points (the CPU stand-in of the tests) takes the one-call form.
        if not hasattr(engine, "forward_set_lineal"):
            b = None
            if np.ndim(self.b) > 0 or self.b != 0:
                b = np.broadcast_to(np.asarray(self.b, dtype=np.float64), (self.n_obs,))
            return engine.forward_lineal(self.A, U_dev, b=b, out=out)
        self.ensure_installed(engine)
        return engine.forward_apply(U_dev, out=out)

    def ensure_installed(self, engine):
        """Make THIS map the one installed in ``engine`` (build-only).  Called by ``forward_device`` and -- before the moment
        kernels that read the installed map, cesx_moments_rest_lineal -- by ``ShardedUpdate.begin_lineal``: the G-dependent
        moments, G itself and K3 then all use the same map in every step (an edit of A or b between two steps, or another
        model that installed its map on the same engine, re-installs first)."""
        # the installed map is reused while A and b are THE SAME OBJECTS (strong references are kept, so an id cannot be
        # recycled) and a cheap fingerprint of their contents (shape, dtype, 64 strided samples, the sum) is unchanged:
        # an in-place edit of A or b between two calls re-installs the map.  ``invalidate_device()`` forces it.
        fp = self._fingerprint()
        if (getattr(self, "_dev_A", None) is not self.A or getattr(self, "_dev_b", None) is not self.b
                or getattr(self, "_dev_fp", None) != fp
                or getattr(engine, "_fwd_token", None) is not getattr(self, "_dev_token", 0)):
            b = None
            if np.ndim(self.b) > 0 or self.b != 0:
                b = np.broadcast_to(np.asarray(self.b, dtype=np.float64), (self.n_obs,))
            self._dev_token = engine.forward_set_lineal(np.asarray(self.A), b)      # (another model may have installed its map)
            self._dev_A, self._dev_b, self._dev_fp = self.A, self.b, fp


def __getattr__(name):
    # re-export ces_amd.models lazily (it imports ``lineal`` from this module)
    from . import models
    try:
        return getattr(models, name)
    except AttributeError:
        raise AttributeError("module %r has no attribute %r" % (__name__, name)) from None


def hook_takes_out(hook):
    """Whether a model's ``forward_device(engine, U[, out=])`` hook accepts ``out=`` -- decided from its signature, once, not by
    catching ``TypeError`` around the call (a ``TypeError`` raised INSIDE a hook that does take ``out=`` would be swallowed and the
    forward map evaluated a second time through the copy fallback)."""
    import inspect
    try:
        prm = inspect.signature(hook).parameters
    except (TypeError, ValueError):
        return False
    return "out" in prm or any(q.kind is inspect.Parameter.VAR_KEYWORD for q in prm.values())
