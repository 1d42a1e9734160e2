"""ctypes binding of ``libcesx.so`` (include/cesx.h).

PyTorch-ROCm tensors are used only as device buffers and for the current HIP
stream; every number is produced by the hand-written HIP kernels behind the C
ABI.  There is no CPU fallback: if the library is missing or no GPU is
present the product path raises.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CESX_LIB") or os.path.join(_HERE, "libcesx.so")      # (CESX_LIB: a dev build of the same ABI)

OK, EINVAL, ENOTPD, EHIP, ESTATE, EUNSUPPORTED, ENOCONV = 0, 1, 2, 3, 4, 5, 6
F32, F64 = 0, 1
UPDATES = {"eks": 0, "aldi": 1, "aldi_constant": 2}
TIME_STEPS = {None: 0, "spectral": 1, "constant": 2, "adaptive": 3, "mix": 4}
ABI_VERSION = 1

EXPORTS = ("cesx_abi_version", "cesx_create", "cesx_destroy", "cesx_last_error", "cesx_set_problem",
           "cesx_step", "cesx_result", "cesx_moments_len", "cesx_colsum", "cesx_set_shift",
           "cesx_moments", "cesx_apply", "cesx_apply_drift", "cesx_apply_finish", "cesx_draw_noise",
           "cesx_forward_lineal", "cesx_debug_dense", "cesx_profile_enable", "cesx_profile_read",
           "cesx_moments_uu_len", "cesx_moments_uu", "cesx_chol_async", "cesx_moments_rest", "cesx_side_stream",
           "cesx_prefetch_noise", "cesx_forward_set_lineal", "cesx_forward_apply", "cesx_moments_uu_chol", "cesx_debug_poll_recoveries", "cesx_comm_unique_id", "cesx_comm_init", "cesx_comm_destroy", "cesx_comm_nranks",
           "cesx_comm_stats", "cesx_allreduce_head", "cesx_allreduce_tail", "cesx_allreduce_whole", "cesx_allreduce_sum", "cesx_allreduce_max", "cesx_moments_uu_handover", "cesx_debug_gram_plan",
           "cesx_profile_clock", "cesx_calibrate_mfma", "cesx_profile_gap", "cesx_moments_rest_lineal", "cesx_copy_cols_async",
           "cesx_debug_warm_inverse", "cesx_debug_update_form", "cesx_comm_count")


class Config(C.Structure):
    _fields_ = [("struct_bytes", C.c_uint32), ("p", C.c_int32), ("n_obs", C.c_int32), ("dtype", C.c_int32),
                ("device", C.c_int32), ("J_local", C.c_int64), ("J_global", C.c_int64),
                ("j_offset", C.c_int64), ("seed", C.c_uint64)]


class StepParams(C.Structure):
    _fields_ = [("struct_bytes", C.c_uint32), ("update", C.c_int32), ("time_step", C.c_int32),
                ("first_step", C.c_int32), ("t_len", C.c_int32), ("reserved", C.c_int32),
                ("t_last", C.c_double), ("delta_t", C.c_double), ("spinup", C.c_double),
                ("switch_mult", C.c_double), ("step_index", C.c_uint64)]


class StepResult(C.Structure):
    _fields_ = [("hk", C.c_double), ("t_new", C.c_double), ("self_bias", C.c_double),
                ("self_bias_data", C.c_double), ("bias_data", C.c_double), ("bias", C.c_double),
                ("radspec", C.c_double), ("lag_bias_data", C.c_double), ("lag_self_bias_data", C.c_double),
                ("status", C.c_int32), ("reserved", C.c_int32)]


class CesxError(RuntimeError):
    def __init__(self, code, msg, result=None):
        super().__init__("cesx error %d: %s" % (code, msg))
        self.code = code
        self.result = result      # cesx_result's CESX_ESTATE after a re-run step: the (valid) result of that step


_lib = None


def cpu_share():
    """Host cores this process may actually use: the cgroup CPU quota when there is one (a GPU box exposes 256
    logical CPUs and gives a one-GPU job a 16-core share), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def default_copy_threads():
    """Host threads for the staging casts.  They are memory-bound (4 threads already move a 256 x 65 536 block in
    1.5 ms) and they run right after the caller's own BLAS threads, which keep spinning for a while: on a box with
    a cgroup CPU quota the two pools TOGETHER must stay within the quota, or the kernel throttles the whole process
    for the rest of the 100-ms period -- 40-50 ms stalls in whatever phase runs next (round 3, tools/hostloop_probe.py:
    16 BLAS + 16 copy threads on a 16-core share: every period throttled, 30.7 ms per iteration; 12 + 4: none, 18.5).
    A quarter of the share, at least 2, at most 8 (CESX_COPY_THREADS overrides)."""
    env = os.environ.get("CESX_COPY_THREADS")
    if env:
        return max(1, int(env))
    return max(2, min(8, cpu_share() // 4))


def load_library(path=None):
    """Load libcesx.so and declare its prototypes.  Loud failure when absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `python -m ces_amd.build` "
                          "(there is no CPU fallback for the HIP engine)" % path)
    lib = C.CDLL(path)
    vp, i32, u64, dp = C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_double)
    lib.cesx_abi_version.restype = C.c_int
    lib.cesx_last_error.restype = C.c_char_p
    lib.cesx_last_error.argtypes = [vp]
    lib.cesx_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.cesx_destroy.argtypes = [vp]
    lib.cesx_destroy.restype = None
    lib.cesx_set_problem.argtypes = [vp, dp, dp, dp, dp, dp]
    lib.cesx_step.argtypes = [vp, C.POINTER(StepParams), vp, vp, vp, vp, i32, vp]
    lib.cesx_result.argtypes = [vp, C.POINTER(StepResult)]
    lib.cesx_moments_len.argtypes = [vp]
    lib.cesx_moments_len.restype = C.c_size_t
    lib.cesx_colsum.argtypes = [vp, vp, vp, vp, vp]
    lib.cesx_set_shift.argtypes = [vp, vp, vp]
    lib.cesx_moments.argtypes = [vp, vp, vp, vp, vp]
    lib.cesx_moments_uu_len.argtypes = [vp]
    lib.cesx_moments_uu_len.restype = C.c_size_t
    lib.cesx_moments_uu.argtypes = [vp, vp, vp, vp, vp]
    lib.cesx_moments_rest.argtypes = [vp, vp, vp, vp, vp]
    lib.cesx_moments_rest_lineal.argtypes = [vp, vp, vp]
    lib.cesx_chol_async.argtypes = [vp, i32, vp, vp]
    lib.cesx_side_stream.argtypes = [vp]
    lib.cesx_side_stream.restype = vp
    lib.cesx_apply.argtypes = [vp, C.POINTER(StepParams), vp, vp, vp, vp, vp, vp]
    lib.cesx_apply_drift.argtypes = [vp, C.POINTER(StepParams), vp, vp, vp, vp, vp, vp]
    lib.cesx_apply_finish.argtypes = [vp, C.POINTER(StepParams), vp, vp, vp, vp, vp]
    lib.cesx_draw_noise.argtypes = [vp, u64, vp, vp]
    lib.cesx_prefetch_noise.argtypes = [vp, u64, vp]
    lib.cesx_moments_uu_chol.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
    lib.cesx_moments_uu_handover.argtypes = [vp, vp, vp, vp, vp]
    lib.cesx_comm_unique_id.argtypes = [vp]
    lib.cesx_comm_init.argtypes = [vp, i32, i32, vp]
    lib.cesx_comm_destroy.argtypes = [vp]
    lib.cesx_comm_nranks.argtypes = [vp]
    lib.cesx_comm_count.argtypes = [vp]
    lib.cesx_comm_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    for name in ("cesx_allreduce_head", "cesx_allreduce_tail", "cesx_allreduce_whole"):
        getattr(lib, name).argtypes = [vp, vp, vp]
    for name in ("cesx_allreduce_sum", "cesx_allreduce_max"):
        getattr(lib, name).argtypes = [vp, vp, C.c_size_t, vp]
    lib.cesx_debug_poll_recoveries.argtypes = [vp]
    lib.cesx_debug_poll_recoveries.restype = C.c_ulonglong
    lib.cesx_debug_warm_inverse.argtypes = [vp]
    lib.cesx_debug_update_form.argtypes = [vp]
    lib.cesx_forward_lineal.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.cesx_forward_set_lineal.argtypes = [vp, vp, vp, vp]
    lib.cesx_forward_apply.argtypes = [vp, vp, vp, vp]
    lib.cesx_debug_dense.argtypes = [vp, dp, dp, dp, dp, dp, dp]
    lib.cesx_debug_gram_plan.argtypes = [i32, i32, i32, i32, i32, C.c_longlong, C.POINTER(C.c_int)]
    lib.cesx_profile_enable.argtypes = [vp, i32]
    lib.cesx_profile_read.argtypes = [vp, i32, dp, C.POINTER(C.c_int)]
    lib.cesx_profile_clock.argtypes = [vp, dp]
    lib.cesx_profile_gap.argtypes = [vp, dp]
    lib.cesx_copy_cols_async.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, i32, vp]
    lib.cesx_calibrate_mfma.argtypes = [vp, C.c_double, dp, dp, vp]
    if lib.cesx_abi_version() != ABI_VERSION:
        raise ImportError("libcesx.so ABI %d != binding ABI %d" % (lib.cesx_abi_version(), ABI_VERSION))
    if path == LIB_PATH:
        _lib = lib
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def step_params(update="aldi", time_step=None, first_step=True, t_len=0, t_last=0.0, delta_t=None,
                spinup=4.0, switch=1.0, step_index=0, T=30):
    """kwargs of sampling.eks_update* (ces/calibrate.py:247-260, :517) -> cesx_step_params."""
    if update not in UPDATES:
        raise ValueError("unknown update rule %r" % (update,))
    if time_step not in TIME_STEPS:
        # the reference leaves hk unbound for unknown strings (ces/calibrate.py:262)
        raise UnboundLocalError("local variable 'hk' referenced before assignment")
    prm = StepParams()
    prm.struct_bytes = C.sizeof(StepParams)
    prm.update = UPDATES[update]
    prm.time_step = TIME_STEPS[time_step]
    prm.first_step = int(bool(first_step))
    prm.t_len = int(t_len)
    prm.t_last = float(t_last)
    prm.delta_t = float(delta_t if delta_t is not None else 1.0 / (T / 2))
    prm.spinup = float(spinup)
    prm.switch_mult = float(switch)
    prm.step_index = int(step_index)
    return prm


class Engine:
    """One handle = one device = one particle shard (cesx_create)."""

    def __init__(self, p, n_obs, J, dtype="float32", device=0, J_global=None, j_offset=0, seed=1234):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("ces_amd needs a HIP device: the ensemble update has no CPU path")
        self.p, self.n_obs, self.J = int(p), int(n_obs), int(J)
        self.J_global = int(J_global if J_global is not None else J)
        self.np_dtype = np.dtype(dtype)
        if self.np_dtype not in (np.dtype("float32"), np.dtype("float64")):
            raise ValueError("dtype must be float32 or float64")
        self.torch_dtype = torch.float32 if self.np_dtype == np.dtype("float32") else torch.float64
        self.device = torch.device("cuda", int(device))
        self._dev_index = int(device)
        cfg = Config()
        cfg.struct_bytes = C.sizeof(Config)
        cfg.p, cfg.n_obs = self.p, self.n_obs
        cfg.dtype = F32 if self.np_dtype == np.dtype("float32") else F64
        cfg.device = int(device)
        cfg.J_local, cfg.J_global, cfg.j_offset = self.J, self.J_global, int(j_offset)
        cfg.seed = int(seed)
        self._h = C.c_void_p()
        rc = self.lib.cesx_create(C.byref(cfg), C.byref(self._h))
        if rc != OK:
            raise CesxError(rc, self.lib.cesx_last_error(None).decode())
        self._problem = None

    def __del__(self):
        pool = self.__dict__.pop("_out_pool", None)
        if pool is not None:
            pool.close()
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.lib.cesx_destroy(h)

    # -- helpers ---------------------------------------------------------
    def _check(self, rc):
        if rc == OK:
            return
        msg = self.lib.cesx_last_error(self._h).decode()
        if rc == ENOTPD:
            raise np.linalg.LinAlgError(msg or "Matrix is not positive definite")
        if rc == ENOCONV:                 # what np.linalg.eigvals (ces/calibrate.py:250) raises
            raise np.linalg.LinAlgError("Eigenvalues did not converge: " + msg)
        if rc == EUNSUPPORTED:
            raise AttributeError("'sampling' object has no attribute 'LM_procedure'")
        if rc == EINVAL:
            raise ValueError(msg)
        raise CesxError(rc, msg)

    def _stream(self):
        # (the raw handle of torch's current stream: ~0.3 us through the C binding, 2 us through the Stream object --
        #  a step of a small problem is bound by the driving thread, tools/host_call_cost.py)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return C.c_void_p(raw(self._dev_index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # Host arrays cross PCIe through persistent pinned staging buffers in the ENGINE dtype: the
    # cast float64 -> engine dtype is a multi-threaded torch copy into the pinned buffer (1.4 ms
    # for a 256 x 65 536 block, against 11 ms for numpy astype + a pageable upload), the way back
    # is pinned engine-dtype -> float64 on the host (3.9 ms against 32 ms).  Large blocks only.
    _PIN_MIN = 1 << 16         # from here on: multi-threaded torch casts, chunked D2H into pre-faulted result arrays
    _PIN_SMALL = 1 << 10       # ... and from here on pinned staging at all (numpy's own cast; the reference's sizes, J <= 768)
    copy_threads = default_copy_threads()     # host threads for the staging casts (see default_copy_threads)

    class _HostThreads:
        """Bound torch's intra-op thread count for a host-side copy (a 128-thread pool on a
        16-core CPU share turned 2 ms casts into 90 ms ones)."""

        def __init__(self, n):
            self.n = n

        def __enter__(self):
            self.old = torch.get_num_threads()
            if self.old > self.n:
                torch.set_num_threads(self.n)

        def __exit__(self, *exc):
            if torch.get_num_threads() != self.old:
                torch.set_num_threads(self.old)

    def _pinned(self, tag, shape):
        pool = self.__dict__.setdefault("_pin_pool", {})
        key = (tag, tuple(shape))
        if key not in pool:
            try:
                pool[key] = torch.empty(shape, dtype=self.torch_dtype).pin_memory()
            except RuntimeError:
                pool[key] = None                         # no pinned memory left: pageable path
        return pool[key]

    def to_device(self, a, rows=None, tag="in"):
        """Host (rows, J) array or device tensor -> contiguous device tensor of the engine dtype."""
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=self.torch_dtype).contiguous()
        else:
            a = np.asarray(a)
            pin = self._pinned(tag, a.shape) if a.size >= self._PIN_SMALL and a.dtype.kind == "f" else None
            if pin is not None and a.size < self._PIN_MIN:
                # small arrays: cast straight into the pinned buffer with numpy (no torch thread-pool juggling, no pageable
                # staging copy inside the runtime: ~10 us instead of ~25 per array), asynchronous H2D out of it
                evs = self.__dict__.setdefault("_pin_events", {})
                key = (tag, tuple(a.shape))
                if key in evs:
                    evs[key].synchronize()
                views = self.__dict__.setdefault("_pin_views", {})
                view = views.get(key)
                if view is None:
                    view = views[key] = pin.numpy()
                np.copyto(view, a, casting="unsafe")
                with torch.cuda.device(self.device):
                    t = pin.to(self.device, non_blocking=True)
                    ev = evs.get(key)
                    if ev is None:
                        ev = evs[key] = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
            elif pin is not None:
                # Asynchronous H2D out of the tag's own staging buffer: the cast of the NEXT array (another tag)
                # runs on the host while this one crosses PCIe.  The buffer is written again only after the event
                # of its last copy has completed.
                evs = self.__dict__.setdefault("_pin_events", {})
                key = (tag, tuple(a.shape))
                if key in evs:
                    evs[key].synchronize()
                with self._HostThreads(self.copy_threads):
                    pin.copy_(torch.from_numpy(np.ascontiguousarray(a)))
                with torch.cuda.device(self.device):
                    t = pin.to(self.device, non_blocking=True)
                    ev = evs.get(key)
                    if ev is None:
                        ev = evs[key] = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
            else:
                t = torch.as_tensor(np.ascontiguousarray(a, dtype=self.np_dtype), device=self.device)
        if rows is not None and tuple(t.shape) != (rows, self.J):
            raise ValueError("expected shape (%d, %d), got %s" % (rows, self.J, tuple(t.shape)))
        return t

    class _HostOutPool:
        """Fresh float64 result arrays, page-faulted AHEAD of use by a helper thread.

        The reference returns a newly allocated (p, J) float64 array from every update
        (ces/calibrate.py:443/:484/:525 build ``Uk`` from temporaries) and the trace keeps it alive, so
        the drop-in must hand out fresh memory too -- and at J = 65 536, p = 256 the first touch of 134 MB
        of new pages costs 12-20 ms, more than everything else in the call together.  The pool keeps a
        few arrays per shape allocated and touched by a background thread; ``get`` is then a list pop.
        One ready list per shape (a trace of U (p, J) and G (n, J) alternates two shapes); the two most recently
        asked-for shapes are kept, older ones are dropped on the helper thread.  All counters are guarded by one
        condition variable; ``get`` never waits longer than ``wait_s`` for the helper (it allocates itself then)."""

        keep_shapes = 2
        wait_s = 2.0

        def __init__(self, depth=2):
            import queue
            import threading
            self.depth = depth
            self.shape = None                     # the shape asked for last
            self.recycled = 0
            self.cv = threading.Condition()
            self.ready = {}                       # shape -> arrays with their pages mapped (insertion order = LRU)
            self.pending = {}                     # shape -> fresh arrays asked for and not yet delivered
            self.req = queue.Queue()
            self.th = threading.Thread(target=self._work, daemon=True, name="cesx-hostpool")
            self.th.start()

        def close(self):
            """Stop the helper thread and release the prepared arrays (Engine.__del__)."""
            self.req.put(None)
            with self.cv:
                self.ready.clear()
                self.pending.clear()

        _touch_pool = None

        @classmethod
        def _fresh(cls, shape, threads=8):
            a = np.empty(shape, dtype=np.float64)
            flat = a.reshape(-1)
            n = flat.size
            if n < (1 << 22):
                flat[::512] = 0.0                   # one write per 4 KiB page: the kernel maps (and zeroes) it now
                return a
            # the kernel zeroes every new page (~10 GB/s per faulting thread): touch slices from several threads
            if cls._touch_pool is None:
                from concurrent.futures import ThreadPoolExecutor
                cls._touch_pool = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="cesx-prefault")
            step = (n // threads + 511) // 512 * 512

            def touch(k):
                flat[k * step:min(n, (k + 1) * step):512] = 0.0
            list(cls._touch_pool.map(touch, range((n + step - 1) // step)))
            return a

        def _work(self):
            while True:
                item = self.req.get()
                if item is None:
                    return
                if isinstance(item, list):           # arrays handed over by discard()
                    self._take_back(item)
                    continue
                with self.cv:
                    wanted = item in self.ready       # (the shape may have been evicted since it was asked for)
                a = self._fresh(item) if wanted else None
                with self.cv:
                    self.pending[item] = max(0, self.pending.get(item, 0) - 1)
                    if a is not None and item in self.ready:
                        self.ready[item].append(a)
                    self.cv.notify_all()
                del a

        def _take_back(self, arrays):
            """Arrays the caller is done with.  One that nobody else refers to, of a shape being handed out, goes
            back into that shape's ready list -- its pages are mapped, the next ``get`` costs neither a page fault
            nor, here, an ``munmap`` (together 16-20 ms per 134 MB array and iteration); anything else is dropped
            (unmapped) on this thread.  The caller drops its own references right after ``discard`` returns: wait
            for that."""
            import sys
            import time
            while arrays:
                a = arrays.pop()
                ok = isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.owndata and a.flags.c_contiguous
                if ok:
                    with self.cv:
                        lst = self.ready.get(tuple(a.shape))
                        ok = lst is not None and len(lst) <= self.depth
                if ok:
                    for _ in range(20):                      # holders: `a` and getrefcount's argument
                        if sys.getrefcount(a) <= 2:
                            break
                        time.sleep(0.0005)
                    ok = sys.getrefcount(a) <= 2
                if ok:
                    with self.cv:
                        lst = self.ready.get(tuple(a.shape))
                        if lst is not None:
                            lst.append(a)
                            self.recycled += 1
                            self.cv.notify_all()
                del a

        def discard(self, arrays):
            """Drop large host arrays on the helper thread: unmapping 134 MB takes ~8 ms on the caller's
            thread otherwise (the caller must not keep a reference of its own)."""
            if self.th.is_alive():
                self.req.put(list(arrays))

        def _top_up(self, shape):
            """(lock held) ask the helper for fresh arrays until ready + pending reaches the depth."""
            want = self.depth - len(self.ready[shape]) - self.pending.get(shape, 0)
            for _ in range(max(0, want)):
                self.pending[shape] = self.pending.get(shape, 0) + 1
                self.req.put(shape)

        def get(self, shape):
            import time
            shape = tuple(int(x) for x in shape)
            if not self.th.is_alive():
                return self._fresh(shape)
            dropped = []
            with self.cv:
                self.shape = shape
                lst = self.ready.pop(shape, None)
                self.ready[shape] = lst if lst is not None else []        # (re-inserted: most recently used)
                while len(self.ready) > self.keep_shapes:                  # evict the least recently used shape
                    old = next(iter(self.ready))
                    dropped.extend(self.ready.pop(old))
                    self.pending.pop(old, None)
                self._top_up(shape)
                deadline = time.monotonic() + self.wait_s
                while not self.ready[shape]:
                    left = deadline - time.monotonic()
                    if left <= 0 or not self.th.is_alive():
                        break
                    self.cv.wait(min(left, 0.05))
                a = self.ready[shape].pop() if self.ready[shape] else None
                self._top_up(shape)
            if dropped:
                self.req.put(dropped)                                      # unmapped on the helper thread
            return a if a is not None else self._fresh(shape)

    def _host_pool(self):
        """The engine's one result-array pool (created on first use; its helper thread ends with the engine)."""
        pool = self.__dict__.get("_out_pool")
        if pool is None:
            pool = self._out_pool = self._HostOutPool()
        return pool

    def to_host(self, t):
        """Device tensor of the engine dtype -> NEW float64 numpy array (the reference's dtype)."""
        pin = self._pinned("out", t.shape) if t.numel() >= self._PIN_SMALL else None
        if pin is None:
            return t.to("cpu", dtype=torch.float64).numpy()
        if t.numel() < self._PIN_MIN:
            # small: one asynchronous D2H into pinned memory, then numpy's own widening into a fresh array
            evs = self.__dict__.setdefault("_out_events", [])
            with torch.cuda.device(self.device):
                pin.copy_(t, non_blocking=True)
                if not evs:
                    evs.append(torch.cuda.Event())
                evs[0].record(torch.cuda.current_stream(self.device))
            evs[0].synchronize()
            return pin.numpy().astype(np.float64)
        # D2H into pinned memory in row chunks, each widened into the result array (pages already mapped) while the
        # next one is still crossing PCIe
        rows = int(t.shape[0])
        nchunk = 4 if rows >= 8 else 1
        step = (rows + nchunk - 1) // nchunk
        evs = self.__dict__.setdefault("_out_events", [])
        parts = []
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device)
            for k in range(nchunk):
                sl = slice(k * step, min(rows, (k + 1) * step))
                if sl.start >= sl.stop:
                    break
                pin[sl].copy_(t[sl], non_blocking=True)
                if len(evs) <= k:
                    evs.append(torch.cuda.Event())
                evs[k].record(st)
                parts.append((sl, evs[k]))
        out = self._host_pool().get(tuple(t.shape))
        dst = torch.from_numpy(out)
        with self._HostThreads(self.copy_threads):
            for sl, ev in parts:
                ev.synchronize()
                dst[sl].copy_(pin[sl])
        return out

    def discard_host(self, *arrays):
        """Hand large host arrays the caller no longer needs to the helper thread for release."""
        self._host_pool().discard([a for a in arrays if isinstance(a, np.ndarray) and a.nbytes >= (1 << 22)])

    def copy_cols_async(self, dst, src, a, b, to_device, stream=None):
        """Columns [a, b) of a row-major 2-D tensor between a PINNED host tensor and a device tensor of the same shape,
        asynchronously on ``stream`` (cesx_copy_cols_async; torch's own non_blocking copy of a strided block blocks the
        host for the whole transfer)."""
        esz = dst.element_size()
        rows = int(dst.shape[0])
        st = self._stream() if stream is None else C.c_void_p(stream)
        self._check(self.lib.cesx_copy_cols_async(self._h, dst.data_ptr() + a * esz, int(dst.stride(0)) * esz,
                                               src.data_ptr() + a * esz, int(src.stride(0)) * esz,
                                               (b - a) * esz, rows, int(bool(to_device)), st))

    def empty(self, rows):
        return torch.empty((rows, self.J), dtype=self.torch_dtype, device=self.device)

    # -- problem ---------------------------------------------------------
    def set_problem(self, y, Gamma, mu, sigma, ustar):
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(self.n_obs))
        Gamma = np.ascontiguousarray(np.asarray(Gamma, dtype=np.float64).reshape(self.n_obs, self.n_obs))
        mu = np.ascontiguousarray(np.asarray(mu, dtype=np.float64).reshape(self.p))
        sigma = np.asarray(sigma, dtype=np.float64)
        if sigma.ndim == 0:
            sigma = float(sigma) * np.eye(self.p)
        sigma = np.ascontiguousarray(sigma.reshape(self.p, self.p))
        ustar = np.ascontiguousarray(np.asarray(ustar, dtype=np.float64).reshape(self.p))
        key = (y, Gamma, mu, sigma, ustar)
        if self._problem is not None and all(np.array_equal(a, b) for a, b in zip(key, self._problem)):
            return
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_set_problem(self._h, _dptr(y), _dptr(Gamma), _dptr(mu), _dptr(sigma),
                                                  _dptr(ustar)))
        self._problem = tuple(a.copy() for a in key)
        # a dense Gamma: the engine whitens the data once per step and runs the diagonal path (include/cesx.h)
        self.dense_gamma = bool(np.count_nonzero(Gamma - np.diag(np.diagonal(Gamma))))

    # -- single device step ------------------------------------------------
    def step(self, prm, U, G, xi=None, out=None, recenter=True):
        """cesx_step: returns the new (p, J) device tensor (never aliases U)."""
        U, G = self.to_device(U, self.p, "U"), self.to_device(G, self.n_obs, "G")
        xi_t = None if xi is None else self.to_device(xi, self.p, "xi")
        out = self.empty(self.p) if out is None else out
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_step(self._h, C.byref(prm), U.data_ptr(), G.data_ptr(),
                                           None if xi_t is None else xi_t.data_ptr(), out.data_ptr(),
                                           int(bool(recenter)), self._stream()))
        self._keep = (U, G, xi_t, out)      # keep inputs alive until the stream has consumed them
        return out

    def result(self):
        res = StepResult()
        rc = self.lib.cesx_result(self._h, C.byref(res))
        if rc == ESTATE and self.poll_recoveries() > self.__dict__.get("_seen_recoveries", 0):
            # a polled join ran out, the step was re-run and `res` is valid; moments enqueued behind it must be redone
            # (include/cesx.h): the pipelined drivers (ShardedUpdate.result) do that and carry on
            self._seen_recoveries = self.poll_recoveries()
            raise CesxError(rc, self.lib.cesx_last_error(self._h).decode(), result=res)
        self._seen_recoveries = self.poll_recoveries()
        self._check(rc)
        return res

    # -- split entry points -------------------------------------------------
    def moments_len(self):
        return int(self.lib.cesx_moments_len(self._h))

    def colsum(self, U, G):
        sums = torch.empty(1 + self.p + self.n_obs, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_colsum(self._h, U.data_ptr(), G.data_ptr(), sums.data_ptr(), self._stream()))
        return sums

    def set_shift(self, sums):
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_set_shift(self._h, sums.data_ptr(), self._stream()))

    def moments(self, U, G, out=None):
        mom = torch.empty(self.moments_len(), dtype=torch.float64, device=self.device) if out is None else out
        self._check(self.lib.cesx_moments(self._h, U.data_ptr(), G.data_ptr(), mom.data_ptr(), self._stream()))
        return mom

    def moments_uu_len(self):
        return int(self.lib.cesx_moments_uu_len(self._h))

    def side_stream(self):
        """The engine's side stream as a torch stream (cesx_side_stream)."""
        return torch.cuda.ExternalStream(int(self.lib.cesx_side_stream(self._h)), device=self.device)

    def moments_uu(self, U, G, out=None):
        """U x U part of the moments into the leading moments_uu_len() entries of the buffer."""
        mom = torch.empty(self.moments_len(), dtype=torch.float64, device=self.device) if out is None else out
        self._check(self.lib.cesx_moments_uu(self._h, U.data_ptr(), G.data_ptr(), mom.data_ptr(), self._stream()))
        return mom

    def chol_async(self, prm, mom):
        """C = cov(U) and L = chol(C) on the engine's side stream (joined by apply)."""
        self._check(self.lib.cesx_chol_async(self._h, int(prm.update), mom.data_ptr(), self._stream()))

    def moments_uu_chol(self, prm, U, G, out=None):
        """moments_uu + chol_async in one call (one device, nothing between the two: cesx_moments_uu_chol)."""
        mom = torch.empty(self.moments_len(), dtype=torch.float64, device=self.device) if out is None else out
        self._check(self.lib.cesx_moments_uu_chol(self._h, int(prm.update), U.data_ptr(), G.data_ptr(),
                                                  mom.data_ptr(), self._stream()))
        return mom

    # -- the exchange step of a sharded ensemble behind the C ABI (RCCL; include/cesx.h cesx_comm_*) --
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        """Rank 0: the 128-byte id every rank passes to ``comm_init`` (ncclGetUniqueId)."""
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        rc = self.lib.cesx_comm_unique_id(buf)
        if rc != OK:
            raise CesxError(rc, "cesx_comm_unique_id failed (librccl.so not loadable?)")
        return bytes(buf.raw)

    def comm_init(self, nranks, rank, unique_id):
        """Collective over the ranks: an RCCL communicator on this engine's device (ncclCommInitRank)."""
        assert len(unique_id) == self.COMM_ID_BYTES
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_comm_init(self._h, int(nranks), int(rank), C.c_char_p(unique_id)))

    def comm_destroy(self):
        self._check(self.lib.cesx_comm_destroy(self._h))

    def comm_nranks(self):
        return int(self.lib.cesx_comm_nranks(self._h))

    def comm_count(self):
        """Communicators the engine holds: 2 = one per stream (main + side), 1 = shared, 0 = none (cesx_comm_count)."""
        return int(self.lib.cesx_comm_count(self._h))

    def comm_stats(self):
        """(all-reduces issued through the engine's communicator so far, their total payload in doubles)."""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self.lib.cesx_comm_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def allreduce(self, t, op="sum", part=None):
        """In-place RCCL all-reduce on the current stream through the engine's communicator.  ``part`` in
        {"head", "tail", "whole"}: ``t`` is the whole moment buffer and the named piece of it is summed
        (cesx_allreduce_head / _tail / _whole); otherwise ``t`` is any contiguous float64 device tensor."""
        assert t.dtype == torch.float64 and t.is_contiguous()
        if part is not None:
            fn = {"head": self.lib.cesx_allreduce_head, "tail": self.lib.cesx_allreduce_tail, "whole": self.lib.cesx_allreduce_whole}[part]
            assert t.numel() == self.moments_len()
            self._check(fn(self._h, t.data_ptr(), self._stream()))
        else:
            fn = self.lib.cesx_allreduce_max if op == "max" else self.lib.cesx_allreduce_sum
            self._check(fn(self._h, t.data_ptr(), t.numel(), self._stream()))
        return t

    def warm_inverse(self):
        """1 when the last hk-dependent SPD inverse of a step came from the warm start (cesx_debug_warm_inverse)."""
        return int(self.lib.cesx_debug_warm_inverse(self._h))

    def update_form(self):
        """Form of the update GEMM the last step launched: 0 assembled, 1 hk-free, 2 through the Cholesky factor
        (cesx_debug_update_form)."""
        return int(self.lib.cesx_debug_update_form(self._h))

    def poll_recoveries(self):
        """Steps whose polled join of the side stream ran out and that cesx_result re-ran (include/cesx.h)."""
        return int(self.lib.cesx_debug_poll_recoveries(self._h))

    def moments_uu_handover(self, U, G, out=None):
        """moments_uu on the current stream, then the engine's side stream waits for it (cesx_moments_uu_handover)."""
        mom = torch.empty(self.moments_len(), dtype=torch.float64, device=self.device) if out is None else out
        self._check(self.lib.cesx_moments_uu_handover(self._h, U.data_ptr(), G.data_ptr(), mom.data_ptr(), self._stream()))
        return mom

    def moments_rest(self, U, G, mom):
        self._check(self.lib.cesx_moments_rest(self._h, U.data_ptr(), G.data_ptr(), mom.data_ptr(), self._stream()))
        return mom

    def moments_rest_lineal(self, mom):
        """The G part of the moments from the head of ``mom`` and the installed linear map (cesx_moments_rest_lineal)."""
        self._check(self.lib.cesx_moments_rest_lineal(self._h, mom.data_ptr(), self._stream()))
        return mom

    def apply(self, prm, mom, U, G, xi=None, out=None):
        out = self.empty(self.p) if out is None else out
        self._check(self.lib.cesx_apply(self._h, C.byref(prm), mom.data_ptr(), U.data_ptr(), G.data_ptr(),
                                        None if xi is None else xi.data_ptr(), out.data_ptr(), self._stream()))
        self._keep = (mom, U, G, xi, out)
        return out

    def apply_drift(self, prm, mom, U, G, out):
        absmax = torch.empty(1, dtype=torch.float64, device=self.device)
        self._check(self.lib.cesx_apply_drift(self._h, C.byref(prm), mom.data_ptr(), U.data_ptr(),
                                              G.data_ptr(), out.data_ptr(), absmax.data_ptr(), self._stream()))
        return absmax

    def apply_finish(self, prm, absmax, U, xi, out):
        self._check(self.lib.cesx_apply_finish(self._h, C.byref(prm), absmax.data_ptr(), U.data_ptr(),
                                               None if xi is None else xi.data_ptr(), out.data_ptr(),
                                               self._stream()))
        self._keep = (absmax, U, xi, out)
        return out

    def draw_noise(self, step_index):
        xi = self.empty(self.p)
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_draw_noise(self._h, int(step_index), xi.data_ptr(), self._stream()))
        return xi

    def prefetch_noise(self, step_index):
        """Draw the noise block of ``step_index`` ahead of its update (cesx_prefetch_noise)."""
        self._check(self.lib.cesx_prefetch_noise(self._h, int(step_index), self._stream()))

    def forward_lineal(self, A, U, b=None, out=None):
        A = torch.as_tensor(A).to(device=self.device, dtype=self.torch_dtype).contiguous()
        if tuple(A.shape) != (self.n_obs, self.p):
            raise ValueError("A must be (n_obs, p)")
        bt = None if b is None else torch.as_tensor(b).to(device=self.device, dtype=self.torch_dtype).reshape(-1).contiguous()
        out = self.empty(self.n_obs) if out is None else out
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_forward_lineal(self._h, A.data_ptr(), None if bt is None else bt.data_ptr(),
                                                     U.data_ptr(), out.data_ptr(), self._stream()))
        self._keep = (A, bt, U, out)
        return out

    def forward_set_lineal(self, A, b=None):
        """Install the linear map G = A U + b in the engine (cesx_forward_set_lineal); the engine keeps its own copy."""
        A = torch.as_tensor(A).to(device=self.device, dtype=self.torch_dtype).contiguous()
        if tuple(A.shape) != (self.n_obs, self.p):
            raise ValueError("A must be (n_obs, p)")
        bt = None if b is None else torch.as_tensor(np.ascontiguousarray(b)).to(device=self.device, dtype=self.torch_dtype).reshape(-1).contiguous()
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_forward_set_lineal(self._h, A.data_ptr(), None if bt is None else bt.data_ptr(),
                                                         self._stream()))
            torch.cuda.current_stream(self.device).synchronize()       # A, bt may be released now
        self._fwd_token = object()          # identifies the installed map (ces_amd.utils.lineal checks it)
        return self._fwd_token

    def forward_apply(self, U, out=None):
        """G = A U + b with the installed map (cesx_forward_apply)."""
        out = self.empty(self.n_obs) if out is None else out
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_forward_apply(self._h, U.data_ptr(), out.data_ptr(), self._stream()))
        self._keep_fwd = (U, out)
        return out

    def profile_enable(self, on=True):
        """on: False / True, 2 = bind only the events cesx_profile_gap needs, 3 / 4 = the update / the moments
        launches alone."""
        mode = int(on) if (on is not True and on is not False and on in (2, 3, 4)) else int(bool(on))
        self._check(self.lib.cesx_profile_enable(self._h, mode))

    def profile_read(self, which):
        """(total ms, launches) of kernel 0 = Gram (K1) or 1 = update (K3) since the last read."""
        ms, cnt = C.c_double(), C.c_int()
        self._check(self.lib.cesx_profile_read(self._h, int(which), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def profile_gap(self):
        """ms from the end of the last profiled Gram launch to the start of the last profiled update launch
        (cesx_profile_gap; call before profile_read); None when nothing was profiled."""
        ms = C.c_double()
        self._check(self.lib.cesx_profile_gap(self._h, C.byref(ms)))
        return ms.value if ms.value >= 0 else None

    def profile_clock(self):
        """Shader clock (GHz) of the last profiled update launch (cesx_profile_clock)."""
        ghz = C.c_double()
        self._check(self.lib.cesx_profile_clock(self._h, C.byref(ghz)))
        return ghz.value

    def calibrate_mfma(self, target_ms=5.0):
        """(TFLOP/s, GHz) of a bare MFMA loop of the engine dtype on this device (cesx_calibrate_mfma)."""
        tf, ghz = C.c_double(), C.c_double()
        with torch.cuda.device(self.device):
            self._check(self.lib.cesx_calibrate_mfma(self._h, float(target_ms), C.byref(tf), C.byref(ghz), self._stream()))
        return tf.value, ghz.value

    def debug_dense(self):
        p, n = self.p, self.n_obs
        out = dict(ubar=np.empty(p), gbar=np.empty(n), C=np.empty((p, p)), L=np.empty((p, p)),
                   K=np.empty((p, n)), M=np.empty((p, p)))
        self._check(self.lib.cesx_debug_dense(self._h, _dptr(out["ubar"]), _dptr(out["gbar"]), _dptr(out["C"]),
                                              _dptr(out["L"]), _dptr(out["K"]), _dptr(out["M"])))
        return out
