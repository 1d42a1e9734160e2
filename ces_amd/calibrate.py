"""Host-side mirror of the reference's "Calibrate" interface for the EKS hot path.

Same class names, constructor, attributes, method signatures, kwargs and error
behaviour as ``ces/calibrate.py`` (``enka`` :12-237, ``sampling`` :241-529), so
existing scripts (examples/scripts/darcy-flow.py:43-93) switch by changing the
import.  The arithmetic of one ensemble update runs in the HIP engine
(``ces_amd.engine`` -> ``libcesx.so``); this module only keeps the bookkeeping
the reference keeps on the object (traces, metric lists, pseudo-time).

Build-only extras (not in the reference):
  ``self.engine_dtype``  'float64' (default, parity) or 'float32' (speed)
  ``self.noise``         'numpy'  -- draw xi with np.random.normal(0,1,[p,J]) exactly
                                     like ces/calibrate.py:447/:488/:527, so a seeded
                                     run reproduces the reference's trajectory (default)
                         'device' -- draw xi on the GPU (Philox4x32-10)
  kwarg ``xi=``          inject the noise block of one update
  kwarg ``xis=``         (``run``) inject the noise blocks of a whole run, one (p, J) block per iteration
  ``self.trace_stride``  (``run``, device-resident loop) keep every k-th iterate in ``Uall`` / ``Gall``
                         (default 1 = every iterate, the reference's behaviour; the first and the final
                         ensemble are always kept) -- at J = 65 536, p = 256 a full trace is 268 MB of
                         PCIe traffic and fresh host memory per iteration
  ``run_eks``            alias of ``sampling.run`` (BASELINE.json's name for it)

``sampling.run`` keeps the ensemble ON THE DEVICE for the whole loop when it can: the model offers the
build-only hook ``forward_device(engine, U_dev)`` (``ces_amd.utils.lineal`` does) and the noise does not
have to come from numpy's global stream (``self.noise == 'device'`` or ``xis=`` given).  Per iteration
nothing crosses PCIe but the step's scalars (and the trace, if kept).  Otherwise the reference's data flow
is kept -- host forward map (``G_ens``), float64 numpy arrays into and out of every update.  For LARGE
ensembles with a host ``type == 'map'`` model and device noise that flow is software-pipelined over column
blocks of the ensemble (``_run_host_pipelined``): ``G_ens`` evaluates the particles block by block (it is a
per-particle loop in the reference, ces/calibrate.py:123-130, so the blocks give the same numbers) while the
previous block's G crosses PCIe and the next block of the new ensemble arrives; the ensemble itself is not
uploaded again (the device holds what it produced).  ``self.host_pipeline = False`` switches it off.
"""
import multiprocessing
import os
import pickle

import numpy as np

from . import engine as _engine

try:                                   # progress bars exactly where the reference has them
    from tqdm.autonotebook import tqdm
except Exception:                      # pragma: no cover - tqdm is optional plumbing
    def tqdm(it=None, **_kw):
        return it

_METRIC_KEYS = ("self-bias", "self-bias-data", "bias-data", "bias", "t")


class enka(object):
    """State and forward-map evaluators (ces/calibrate.py:12-237)."""

    def __init__(self, p, n_obs, J):
        # ces/calibrate.py:14-22
        self.n_obs = n_obs
        self.p = p
        self.J = J
        self.epsilon = 1e-7
        self.T = 30
        self.num_cores = multiprocessing.cpu_count()
        self.parallel = False
        self.mute_bar = True
        # build-only
        self.engine_dtype = "float64"
        self.noise = "numpy"
        self.seed = 1234
        self.device = 0

    def __repr__(self):
        # ces/calibrate.py:24-28 (getattr with one argument always raises there)
        return "enka" + "-" + str(self.J).zfill(4) + "-eks"

    def __str__(self):
        # ces/calibrate.py:30-48
        print(r"Number of parameters ................. %s" % (self.p))
        print(r"Dimension of forward model output .... %s" % (self.n_obs))
        print(r"Ensemble size ........................ %s" % (self.J))
        print(r"Evaluate G in parallel ............... %s" % (self.parallel))
        print(r"Number of iterations to be run ....... %s" % (self.T))
        if not hasattr(self, "directory"):
            self.directory = os.getcwd()
        print("Path to save: ......................... %s" % ("~/.../" + "/".join(self.directory.split("/")[-2:])))
        if hasattr(self, "Uall"):
            print(r"Number of iterations EKS has run ..... %s" % (len(self.Uall) - 1))
        else:
            print(r"NOTE: EKS has not been run!")
        return str()

    # -- the base class's placeholders (ces/calibrate.py:50-93): no-ops there, no-ops here; ``sampling`` overrides them --
    def run(self, y_obs, U0, model, Gamma, Jnoise):
        """ces/calibrate.py:50-67: the driver loop's slot in the base class; does nothing, returns None."""
        pass

    def run_sde(self, y_obs, U0, model, Gamma, Jnoise):
        """ces/calibrate.py:69-87: likewise."""
        pass

    def eks_update(self, Geval):
        """ces/calibrate.py:89-93: likewise."""
        pass

    # -- forward map: stays on the host (ces/calibrate.py:95-168) ------------
    def G(self, theta, model):
        return model(theta)

    def G_ens(self, theta, model):
        """ces/calibrate.py:106-130: one model call per particle (serial or joblib)."""
        if self.parallel:
            from joblib import Parallel, delayed
            vals = Parallel(n_jobs=self.num_cores)(delayed(self.G)(k, model) for k in theta.T)
            return np.asarray(vals).T
        Gs = np.zeros((self.n_obs, theta.shape[1]))
        for ii, k in enumerate(theta.T):
            Gs[:, ii] = model(k)
        return Gs

    def G_pde(self, k, model, t):
        """ces/calibrate.py:132-154."""
        w0 = k[self.p:]
        ws = model.solve(w0, t, args=tuple(k[:self.p]))
        gs = model.statistics(ws)
        return np.concatenate([gs, ws[-1]])

    def G_pde_ens(self, theta, model, t):
        """ces/calibrate.py:156-168."""
        if self.parallel:
            from joblib import Parallel, delayed
            vals = Parallel(n_jobs=self.num_cores)(delayed(self.G_pde)(k, model, t) for k in theta.T)
            return np.asarray(vals).T
        Gs = np.zeros((self.n_obs + model.n_state, theta.shape[1]))
        for ii, k in enumerate(theta.T):
            Gs[:, ii] = self.G_pde(k, model, t)
        return Gs

    # -- on-disk format (ces/calibrate.py:170-237) ----------------------------
    def save(self, path="./", file="ces/", all=False, reset=True, online=False, counter=0):
        os.makedirs(path + file, exist_ok=True)
        if not hasattr(self, "Uall"):
            tqdm.write("There is nothing to save") if hasattr(tqdm, "write") else print("There is nothing to save")
            return
        if not online:
            np.save(path + file + "ensemble", self.Ustar)
            np.save(path + file + "Gensemble", self.Gstar)
            with open(path + file + "metrics.pkl", "wb") as fh:
                pickle.dump(self.metrics, fh)
            if all:
                np.save(path + file + "ensemble_path", self.Uall)
                np.save(path + file + "Gensemble_path", self.Gall)
        else:
            np.save(path + file + "ensemble_" + str(counter).zfill(4), self.Uall[-1])
            np.save(path + file + "Gensemble_" + str(counter).zfill(4), self.Gall[-1])
            with open(path + file + "metrics.pkl", "wb") as fh:
                pickle.dump(self.metrics, fh)

    def load(self, path="./", eks_dir="ces/", ix_ensemble=False, flag_metrics=False):
        d = path + eks_dir
        try:
            with open(d + "metrics.pkl", "rb") as fh:
                self.metrics = pickle.load(fh)
        except FileNotFoundError:
            print("Metrics object not found. Could not load EKS object.")
            return False
        if not ix_ensemble:
            try:
                self.Uall = np.load(d + "ensemble_path.npy")
                self.Gall = np.load(d + "Gensemble_path.npy")
            except FileNotFoundError:
                print("EKS trajectory files not found.")
                return False
            return True
        try:
            if flag_metrics:
                count = len(self.metrics["self-bias"])
            else:
                count = int(np.sum([f.split("_")[0] == "ensemble" for f in os.listdir(d)]))
            self.Uall = np.asarray([np.load(d + "ensemble_" + str(i).zfill(4) + ".npy") for i in range(count)])
            self.Gall = np.asarray([np.load(d + "Gensemble_" + str(i).zfill(4) + ".npy") for i in range(count)])
            self.Ustar, self.Gstar = self.Uall[-1], self.Gall[-1]
            self.J = self.Uall.shape[-1]
        except FileNotFoundError:
            return False
        return True


class sampling(enka):
    """EKS / ALDI sampler (ces/calibrate.py:241-529) on the HIP engine."""

    # -- engine plumbing -------------------------------------------------
    def _get_engine(self):
        dtype = str(self.engine_dtype)
        if dtype == "float32" and self.J < 4 * self.p:
            # a rank-deficient ensemble covariance (J - 1 < p) is kept positive definite by the
            # reference's 1e-8 jitter only (ces/calibrate.py:424/:476), and one with few more particles
            # than dimensions is close to that; fp32 moments cannot resolve it, so ensembles smaller
            # than 4 p (cheap anyway) use the fp64 engine
            dtype = "float64"
        key = (self.p, self.n_obs, self.J, dtype, int(self.device), int(self.seed))
        if getattr(self, "_engine_key", None) != key:
            self._engine = _engine.Engine(self.p, self.n_obs, self.J, dtype=dtype,
                                          device=self.device, seed=self.seed)
            self._engine_key = key
            self._step_counter = 0
        return self._engine

    def _is_first_step(self):
        # ces/calibrate.py:262 / :520 test len(self.Uall) == 1; without a trace the
        # reference fails there, the build falls back to "nothing recorded yet"
        if hasattr(self, "Uall"):
            return len(self.Uall) == 1
        return len(self.metrics["t"]) == 0

    def _ensure_metrics(self):
        if not hasattr(self, "metrics"):
            self.radspec = []
            self.metrics = {k: [] for k in _METRIC_KEYS}

    def _device_update(self, rule, y_obs, U0, Geval, Gamma, **kwargs):
        """One ensemble update through libcesx (K1 moments -> K2 dense -> K3 update)."""
        self._ensure_metrics()
        eng = self._get_engine()
        first = self._is_first_step()
        t = self.metrics["t"]
        if not first and len(t) == 0:
            raise IndexError("list index out of range")          # ces/calibrate.py:265 / :523
        prm = _engine.step_params(update=rule, time_step=kwargs.get("time_step", None), first_step=first,
                                  t_len=len(t), t_last=t[-1] if t else 0.0,
                                  delta_t=kwargs.get("delta_t", None), spinup=kwargs.get("spinup", 4.0),
                                  switch=kwargs.get("switch", 1.0), step_index=self._step_counter, T=self.T)
        eng.set_problem(y_obs, Gamma, self.mu, self.sigma, self.ustar)
        if rule != "aldi_constant" and kwargs.get("time_step", None) == "adaptive":
            # ces/calibrate.py:255 calls self.LM_procedure, which is defined nowhere
            raise AttributeError("'sampling' object has no attribute 'LM_procedure'")
        xi = kwargs.get("xi", None)
        if xi is None and self.noise == "numpy":
            xi = np.random.normal(0, 1, [self.p, self.J])         # ces/calibrate.py:447/:488/:527
        # The engine's centring shift follows the ensemble it produced (K2 predicts the next mean): a fresh
        # pass over (U, G) for the shift is only needed for an ensemble the engine has not seen -- the very array
        # the previous update returned, on the same engine, and still holding what was returned (4096 strided
        # samples are compared: a caller that rescales / clips / overwrites the ensemble in place gets a fresh
        # centring pass; the shift only conditions the fp32 moments, the result is exact in either case)
        chained = (U0 is getattr(self, "_last_Uk", None) and eng is getattr(self, "_last_engine", None)
                   and self._sample(U0) == getattr(self, "_last_sample", None))
        try:
            U_next = eng.step(prm, U0, Geval, xi=xi, recenter=not chained)
            res = eng.result()
        finally:
            self._step_counter += 1
        self._append_result(rule, res, kwargs)
        out = eng.to_host(U_next) if isinstance(U0, np.ndarray) else U_next
        self._last_Uk, self._last_engine, self._last_sample = out, eng, self._sample(out)
        return out

    @staticmethod
    def _sample(U):
        """4096 strided entries of a host ensemble as bytes (None for device tensors: nobody edits those in place
        between two updates without going through the engine)."""
        if not isinstance(U, np.ndarray):
            return None
        flat = U.reshape(-1) if U.flags.c_contiguous else None
        if flat is None:
            return b"non-contiguous"
        return flat[::max(1, flat.size // 4096)].tobytes()

    def _append_result(self, rule, res, kwargs):
        """Book-keeping of one update on the object (ces/calibrate.py:432-435, :262-265, :250)."""
        m = self.metrics
        m["self-bias"].append(res.self_bias)
        m["bias"].append(res.bias)
        m["self-bias-data"].append(res.self_bias_data)
        m["bias-data"].append(res.bias_data)
        if rule != "aldi_constant" and kwargs.get("time_step", None) == "spectral":
            self.radspec.append(res.radspec)
        m["t"].append(res.t_new)
        self._last_hk = res.hk

    # -- reference API ---------------------------------------------------
    def timestep_method(self, D, Geval, y_obs, Gamma, Jnoise, **kwargs):
        """Compatibility shim for ces/calibrate.py:243-267 taking a small explicit D.

        The engine never forms D (it takes the step size from n x n moments
        inside ``eks_update*``); this host shim exists for callers that use the
        method on its own.
        """
        self._ensure_metrics()
        rule = kwargs.get("time_step", None)
        if rule is None:
            hk = 1.0 / (np.linalg.norm(D) + 1e-8)
        elif rule == "spectral":
            self.radspec.append(np.linalg.eigvals(D).real.max())
            hk = 1.0 / self.radspec[-1]
        elif rule == "constant":
            hk = kwargs.get("delta_t", 1.0 / (self.T / 2))
        elif rule == "adaptive":
            hk = self.LM_procedure(Geval, y_obs, Gamma, Jnoise, **kwargs)   # undefined, as in the reference
        elif rule == "mix":
            t = self.metrics["t"]
            if len(t) == 0 or t[-1] < kwargs.get("spinup", 4.0):
                hk = 1.0 / (np.linalg.norm(D) + 1e-8)
            else:
                hk = kwargs.get("delta_t", 1.0 / (self.T / 2))
        if len(self.Uall) == 1:
            self.metrics["t"].append(hk)
        else:
            self.metrics["t"].append(hk + self.metrics["t"][-1])
        return hk

    def eks_update(self, y_obs, U0, Geval, Gamma, iter, **kwargs):
        """ces/calibrate.py:418-449."""
        self.update_rule = "eks_update"
        return self._device_update("eks", y_obs, U0, Geval, Gamma, **kwargs)

    def eks_update_aldi(self, y_obs, U0, Geval, Gamma, iter, **kwargs):
        """ces/calibrate.py:451-490."""
        self.update_rule = "eks_update_linear"
        return self._device_update("aldi", y_obs, U0, Geval, Gamma, **kwargs)

    def eks_update_aldi_constant(self, y_obs, U0, Geval, Gamma, iter, **kwargs):
        """ces/calibrate.py:492-529."""
        self.update_rule = "eks_update_aldi"
        return self._device_update("aldi_constant", y_obs, U0, Geval, Gamma, **kwargs)

    def _device_loop_ok(self, model, save_online, kwargs):
        return (getattr(model, "type", None) == "map" and hasattr(model, "forward_device") and not save_online
                and (self.noise == "device" or kwargs.get("xis", None) is not None)
                and kwargs.get("update", "aldi") in _engine.UPDATES
                and kwargs.get("time_step", None) in _engine.TIME_STEPS and kwargs.get("time_step", None) != "adaptive"
                and getattr(self, "device_loop", True))

    def _run_device(self, y_obs, U0, model, Gamma, trace, **kwargs):
        """``run`` with the ensemble resident in HBM (same iteration structure as ces/calibrate.py:341-408):
        forward map through ``model.forward_device``, update through the split entry points of the C ABI,
        software-pipelined like ``ces_amd.dist.ShardedSampler.run`` -- the first half of iteration i+1 (forward
        map, moments, chol(C)) is enqueued BEFORE the host reads t of iteration i for the ``t_tol`` test (:387);
        if the run stops there that work is discarded."""
        from .dist import ShardedUpdate
        rule = kwargs.get("update", "aldi")
        eng = self._get_engine()
        eng.set_problem(y_obs, Gamma, self.mu, self.sigma, self.ustar)
        sh = ShardedUpdate(eng)
        stride = max(1, int(getattr(self, "trace_stride", 1)))
        xis = kwargs.get("xis", None)
        self.update_rule = {"eks": "eks_update", "aldi": "eks_update_linear", "aldi_constant": "eks_update_aldi"}[rule]
        prm0 = _engine.step_params(update=rule, T=self.T)
        U = eng.to_device(U0, self.p, "U")
        G = model.forward_device(eng, U)
        sh.begin(prm0, U, G, recenter=True, noise_step=None if xis is not None else self._step_counter)
        fast = sh.lineal_fast_ok(model)       # linear map on the device: G's moments follow from U's (no second Gram launch)
        G_next = None

        from .utils import hook_takes_out
        takes_out = hook_takes_out(model.forward_device)

        def fwd(u, out=None):                 # (``out``: a redo after a re-run step refreshes G in place, ShardedUpdate.result)
            if out is None:
                return model.forward_device(eng, u)
            if takes_out:
                return model.forward_device(eng, u, out=out)
            out.copy_(model.forward_device(eng, u))          # a hook without ``out=``
            return out
        for i in range(self.T):
            if trace and (i % stride == 0):                        # :356-358 (a copy: the device buffers are reused)
                self.Uall.append(U0 if i == 0 and isinstance(U0, np.ndarray) else eng.to_host(U))
                self.Gall.append(eng.to_host(G))
            t = self.metrics["t"]
            first = len(t) == 0                                    # = len(self.Uall) == 1 of :262 / :520 on a full trace
            prm = _engine.step_params(update=rule, time_step=kwargs.get("time_step", None), first_step=first,
                                      t_len=len(t), t_last=t[-1] if t else 0.0, delta_t=kwargs.get("delta_t", None),
                                      spinup=kwargs.get("spinup", 4.0), switch=kwargs.get("switch", 1.0),
                                      step_index=self._step_counter, T=self.T)
            self._step_counter += 1
            xi = None if xis is None else eng.to_device(xis[i], self.p, "xi")
            U_new = sh.finish(prm, U, G, xi=xi)
            G_next = None
            if i + 1 < self.T:                                     # first half of the next iteration, ahead of the read
                ns = None if xis is not None else self._step_counter
                if fast:
                    _, G_next = sh.begin_lineal(prm0, U_new, fwd, noise_step=ns, model=model)
                else:
                    G_next = model.forward_device(eng, U_new)
                    sh.begin(prm0, U_new, G_next, noise_step=ns, forward=fwd)
            res = sh.result()
            self._append_result(rule, res, kwargs)
            U, G = U_new, G_next
            if self.metrics["t"][-1] > kwargs.get("t_tol", 2.0):   # :387-388
                break
        if G is None:                                              # :390-398 one more evaluation of the final ensemble
            G = model.forward_device(eng, U)
        self.Ustar = eng.to_host(U)
        Gfinal = eng.to_host(G)
        if trace:                                                  # :400-405
            self.Uall.append(self.Ustar)
            self.Gall.append(Gfinal)
            self.Uall = np.asarray(self.Uall)
            self.Gall = np.array(self.Gall)
        self.Gstar = Gfinal[:self.n_obs, :]
        self.Ustar_device, self.Gstar_device = U, G                # build-only: the final ensemble without the D2H copy

    def _updates_overridden(self):
        """A subclass or the instance replaced ``eks_update*`` / ``G_ens`` hooks the pipelined loop would bypass: the
        reference's ``run`` calls those methods (ces/calibrate.py:364-369), so the plain loop -- which does -- runs then."""
        cls = type(self)
        names = ("eks_update", "eks_update_aldi", "eks_update_aldi_constant", "_device_update")
        return any(getattr(cls, nm) is not getattr(sampling, nm) or nm in self.__dict__ for nm in names)

    def _host_pipeline_ok(self, model, save_online, kwargs):
        big = self.p * self.J >= (1 << 22)              # (small ensembles keep the reference's flow call for call)
        return (big and getattr(model, "type", None) == "map" and not save_online and not self._updates_overridden()
                and getattr(self, "host_pipeline", True) and os.environ.get("CESX_HOST_PIPELINE", "1") != "0"
                and (self.noise == "device" or kwargs.get("xis", None) is not None)
                and kwargs.get("update", "aldi") in _engine.UPDATES
                and kwargs.get("time_step", None) in _engine.TIME_STEPS and kwargs.get("time_step", None) != "adaptive")

    def _run_host_pipelined(self, y_obs, U0, model, Gamma, trace, **kwargs):
        """``run`` for a host ``type == 'map'`` model and a large ensemble (same iteration structure as
        ces/calibrate.py:341-408), pipelined over NCH column blocks of the ensemble:

            block c of U_{i+1} arrives (D2H + widening to float64)  ->  G_ens on that block (host)  ->
            its G rows cross PCIe (cast to the engine dtype, H2D)   ...   update of the whole ensemble

        (Side effect, bounded to the duration of the call: torch's intra-op thread pool -- process-wide -- is sized to the
        engine's copy threads while the loop runs, because the staging casts run on it beside the forward map's BLAS
        threads and the two together must not exceed the CPU share; a forward map that itself uses torch CPU ops runs on
        that many threads.  ``self.host_pipeline_torch_threads = 0`` leaves the pool alone, ``host_pipeline = False``
        takes the plain loop.)

        A helper thread does the staging (widening the arriving blocks, casting and sending the G blocks:
        cesx_copy_cols_async, pinned buffers) while this thread evaluates the forward map, so the host forward map of
        one block runs while the blocks around it are on the bus; the ensemble is never uploaded again -- the
        device keeps U_{i+1} as it produced it (the host array is that tensor widened, the same numbers).
        ``G_ens`` is the reference's per-particle loop (:123-130): evaluating it on column blocks gives what one
        call on the whole ensemble gives."""
        import queue
        import threading
        import torch
        from .dist import ShardedUpdate
        rule = kwargs.get("update", "aldi")
        eng = self._get_engine()
        eng.set_problem(y_obs, Gamma, self.mu, self.sigma, self.ustar)
        sh = ShardedUpdate(eng)
        self.update_rule = {"eks": "eks_update", "aldi": "eks_update_linear", "aldi_constant": "eks_update_aldi"}[rule]
        xis = kwargs.get("xis", None)
        p, n, J = self.p, self.n_obs, self.J
        NCH = max(2, min(32, int(os.environ.get("CESX_HOST_PIPE_BLOCKS", getattr(self, "host_pipeline_blocks", 8)))))
        cuts = [(J * c // NCH // 4 * 4, J * (c + 1) // NCH // 4 * 4 if c + 1 < NCH else J) for c in range(NCH)]
        dev, tdt = eng.device, eng.torch_dtype
        pin_g = eng._pinned("hp_g", (n, J))                                             # G blocks on their way up
        pin_u = eng._pinned("hp_u", (p, J))                                             # U_{i+1} on its way down
        if pin_u is None or pin_g is None:
            raise MemoryError("no pinned host memory for the pipelined host loop (set host_pipeline = False)")
        ev_u = [torch.cuda.Event() for _ in range(NCH)]
        ev_g = torch.cuda.Event()
        G_dev = [eng.empty(n), eng.empty(n)]
        U_host = np.ascontiguousarray(np.asarray(U0, dtype=np.float64))
        U_dev = eng.to_device(U_host, p, "U")
        stream = torch.cuda.current_stream(dev)
        raw_stream = int(stream.cuda_stream)
        pool = eng._host_pool()

        # the staging thread: tasks in FIFO order; an exception is handed back to this thread
        tasks, failure = queue.Queue(), []
        import time as _time
        stg = self._pipe_stage_times = {}            # seconds the STAGING thread spent per kind of work (diagnostics, bench.py e2e)

        def stager():
            torch.cuda.set_device(dev)
            while True:
                job = tasks.get()
                if job is None:
                    return
                try:
                    kind, c, arr, done, extra = job
                    a, b = cuts[c]
                    t0 = _time.perf_counter()
                    if kind == "down":               # block c of the new ensemble: wait for its D2H, widen it
                        ev_u[c].synchronize()
                        t1 = _time.perf_counter()
                        torch.from_numpy(arr)[:, a:b].copy_(pin_u[:, a:b])
                        t2 = _time.perf_counter()
                        stg["d2h_wait"] = stg.get("d2h_wait", 0.0) + t1 - t0
                        stg["widen"] = stg.get("widen", 0.0) + t2 - t1
                    else:                            # block c of G: cast into pinned memory, send it up
                        pin_g[:, a:b].copy_(torch.from_numpy(np.asarray(arr)))
                        t1 = _time.perf_counter()
                        eng.copy_cols_async(extra, pin_g, a, b, True, stream=raw_stream)
                        t2 = _time.perf_counter()
                        stg["cast"] = stg.get("cast", 0.0) + t1 - t0
                        stg["h2d_enqueue"] = stg.get("h2d_enqueue", 0.0) + t2 - t1
                except BaseException as ex:          # noqa: B036 -- reported by the driving thread
                    failure.append(ex)
                finally:
                    done.set()
        # torch's intra-op pool serves the staging casts only while this loop runs: sized ONCE to the copy threads
        # (resizing it around every block costs more than the casts)
        old_threads = torch.get_num_threads()
        want_threads = int(getattr(self, "host_pipeline_torch_threads", eng.copy_threads))
        if want_threads > 0 and old_threads > want_threads:
            torch.set_num_threads(want_threads)
        th = threading.Thread(target=stager, daemon=True, name="cesx-hoststage")
        th.start()

        tm = self._pipe_times = {}                   # seconds this thread spent waiting / outside the forward map (diagnostics)

        def wait(evt, what="wait"):
            t0 = _time.perf_counter()
            evt.wait()
            tm[what] = tm.get(what, 0.0) + _time.perf_counter() - t0
            if failure:
                raise failure[0]
        try:
            down = None                              # per-block events of the ensemble on its way down
            for i in range(self.T):
                Gd = G_dev[i % 2]
                G_host = pool.get((n, J)) if trace else None
                ups = []
                for c, (a, b) in enumerate(cuts):
                    if down is not None:
                        wait(down[c], "down%d" % min(c, 1))    # block c of U_host is complete
                    Gc = np.asarray(self.G_ens(U_host[:, a:b], model))[:n, :]
                    if trace:
                        G_host[:, a:b] = Gc
                    if c == 0:
                        ev_g.synchronize()           # (the previous iteration's H2Ds have left the pinned G buffer)
                    done = threading.Event()
                    tasks.put(("up", c, Gc, done, Gd))
                    ups.append(done)
                    del Gc
                for d in ups:
                    wait(d, "up")                    # every G block is cast and its H2D is enqueued on the stream
                ev_g.record(stream)
                down = None
                if trace:                                                  # :356-358
                    self.Uall.append(U_host)
                    self.Gall.append(G_host)
                t = self.metrics["t"]
                prm = _engine.step_params(update=rule, time_step=kwargs.get("time_step", None),
                                          first_step=self._is_first_step(), t_len=len(t), t_last=t[-1] if t else 0.0,
                                          delta_t=kwargs.get("delta_t", None), spinup=kwargs.get("spinup", 4.0),
                                          switch=kwargs.get("switch", 1.0), step_index=self._step_counter, T=self.T)
                self._step_counter += 1
                xi = None if xis is None else eng.to_device(xis[i], p, "xi")
                U_new = sh.step(prm, U_dev, Gd, xi=xi, recenter=(i == 0))
                # the new ensemble starts its way down block by block behind the update kernel; the staging thread
                # widens the blocks as they land, the next iteration's forward map picks them up in order
                U_prev_host = U_host
                U_host = pool.get((p, J))
                down = []
                for c, (a, b) in enumerate(cuts):
                    eng.copy_cols_async(pin_u, U_new, a, b, False, stream=raw_stream)
                    ev_u[c].record(stream)
                    done = threading.Event()
                    tasks.put(("down", c, U_host, done, None))
                    down.append(done)
                t0 = _time.perf_counter()
                res = sh.result()
                tm["result"] = tm.get("result", 0.0) + _time.perf_counter() - t0
                self._append_result(rule, res, kwargs)
                U_dev = U_new
                if not trace and i > 0:
                    eng.discard_host(U_prev_host)    # (iteration 0's array is the caller's U0)
                del U_prev_host
                if self.metrics["t"][-1] > kwargs.get("t_tol", 2.0):       # :387-388
                    break
            if down is not None:
                for d in down:
                    wait(d)
        finally:
            tasks.put(None)
            th.join(timeout=30.0)
            torch.cuda.current_stream(dev).synchronize()
            if torch.get_num_threads() != old_threads:
                torch.set_num_threads(old_threads)
        Geval = self.G_ens(U_host, model)                              # :390-398 the final ensemble, one call
        if trace:                                                      # :400-405
            self.Uall.append(U_host)
            # (a 'map' model returns n_obs rows: every trace entry has them; should one return more, the loop above
            #  kept the first n_obs of each block -- the final entry gets the same rows, so that np.array(Gall) is regular)
            self.Gall.append(np.asarray(Geval)[:self.n_obs, :])
            self.Uall = np.asarray(self.Uall)
            self.Gall = np.array(self.Gall)
        self.Ustar = U_host
        self.Gstar = Geval[:self.n_obs, :]
        self._last_Uk, self._last_engine, self._last_sample = U_host, eng, self._sample(U_host)

    def run(self, y_obs, U0, model, Gamma, Jnoise, save_online=False, trace=True, **kwargs):
        """Driver loop, ces/calibrate.py:270-416."""
        getattr(model, "type")                                     # :294-297
        if not hasattr(self, "directory"):
            self.directory = os.getcwd()
        self.__update = kwargs.get("update", "aldi")               # :304
        if trace:
            if hasattr(self, "Uall"):                              # resume (:307-310)
                self.Uall = list(self.Uall)
                self.Gall = list(self.Gall)
            else:
                self.Uall, self.Gall = [], []
        t = None
        if model.type == "pde":                                    # :317-327
            wt = kwargs.get("wt", None)
            t = kwargs.get("t", None)
            if kwargs.get("ws", None) is not None:
                widx = np.random.randint(kwargs.get("ws").shape[0], size=self.J)
                self.W0 = kwargs.get("ws")[widx].T
                self.Wall = [widx]
            else:
                self.W0 = np.tile(wt, self.J).reshape(self.J, model.n_state).T
        self._ensure_metrics()                                     # :329-339
        if self._device_loop_ok(model, save_online, kwargs):
            self._get_engine()                                     # (creates self._step_counter)
            self._run_device(y_obs, U0, model, Gamma, trace, **kwargs)
            tail = "-" + str(self.nexp).zfill(2) if hasattr(self, "nexp") else ""
            self.online_path = self.directory + "/ensembles/" + model.model_name + "-" + str(self.J).zfill(4) + tail + "/"
            return
        if self._host_pipeline_ok(model, save_online, kwargs) and isinstance(U0, np.ndarray):
            self._get_engine()
            self._run_host_pipelined(y_obs, U0, model, Gamma, trace, **kwargs)
            tail = "-" + str(self.nexp).zfill(2) if hasattr(self, "nexp") else ""
            self.online_path = self.directory + "/ensembles/" + model.model_name + "-" + str(self.J).zfill(4) + tail + "/"
            return

        for i in tqdm(range(self.T), desc="EKS iterations (%s):" % str(self.J), position=1,
                      disable=self.mute_bar):
            if model.type == "pde":                                # :342-350
                Geval = self.G_pde_ens(np.vstack([U0, self.W0]), model, t)
                if kwargs.get("update_wt", True):
                    if kwargs.get("ws", None) is not None:
                        widx = np.random.randint(kwargs.get("ws").shape[0], size=self.J)
                        self.Wall.append(widx)
                        self.W0 = kwargs.get("ws")[widx].T
                    else:
                        self.W0 = np.copy(Geval[self.n_obs:, :])
            elif model.type == "map":
                Geval = self.G_ens(U0, model)
            else:
                break
            if trace:                                              # :356-358
                self.Uall.append(U0)
                self.Gall.append(Geval)
            Geval = Geval[:self.n_obs, :]
            if kwargs.get("xis", None) is not None:                # build-only: injected noise blocks
                kwargs = dict(kwargs, xi=kwargs["xis"][i])
            U_prev = U0
            if self.__update == "eks":                             # :364-369
                U0 = self.eks_update(y_obs, U0, Geval, Gamma, i, **kwargs)
            elif self.__update == "aldi":
                U0 = self.eks_update_aldi(y_obs, U0, Geval, Gamma, i, **kwargs)
            elif self.__update == "aldi_constant":
                U0 = self.eks_update_aldi_constant(y_obs, U0, Geval, Gamma, i, **kwargs)
            if not trace and hasattr(self, "_engine"):
                # nothing else holds the previous iterate or this iteration's forward-map output (no trace; the
                # caller's own U0 is iteration 0): release their pages on the engine's helper thread, not here
                # (unmapping 134 MB costs ~8 ms on the calling thread)
                G_base = Geval.base if isinstance(Geval, np.ndarray) and Geval.base is not None else Geval
                drop = [G_base] + ([U_prev] if i > 0 and U_prev is not U0 else [])
                del Geval, G_base
                self._engine.discard_host(*drop)
                del drop
            del U_prev
            if save_online:                                        # :371-385
                tag = model.model_name + "-eks-" + str(model.l_window).zfill(3) + "-" + str(self.J).zfill(4)
                if hasattr(self, "nexp"):
                    tag += "-" + str(self.nexp).zfill(2)
                self.save(path=self.directory + "/ensembles/", file=tag + "/", online=True, counter=i)
            if self.metrics["t"][-1] > kwargs.get("t_tol", 2.0):   # :387-388
                break

        if model.type == "pde":                                    # :390-398
            Geval = self.G_pde_ens(np.vstack([U0, self.W0]), model, t)
            if kwargs.get("update_wt", True):
                if kwargs.get("ws", None) is not None:
                    self.W0 = kwargs.get("ws")[np.random.randint(kwargs.get("ws").shape[0], size=self.J)].T
                else:
                    self.W0 = Geval[self.n_obs:, :]
        elif model.type == "map":
            Geval = self.G_ens(U0, model)
        if trace:                                                  # :400-405
            self.Uall.append(U0)
            self.Gall.append(Geval)
            self.Uall = np.asarray(self.Uall)
            self.Gall = np.array(self.Gall)
        self.Ustar = U0
        self.Gstar = Geval[:self.n_obs, :]
        tail = "-" + str(self.nexp).zfill(2) if hasattr(self, "nexp") else ""
        self.online_path = self.directory + "/ensembles/" + model.model_name + "-" + str(self.J).zfill(4) + tail + "/"

    run_eks = run
