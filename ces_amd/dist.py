"""Particle-sharded data parallelism for the ensemble update.

One process per GPU.  Rank r owns the column range ``shard_range(J, N, r)`` of
U, G and U_next.  Particles are exchangeable and couple only through first and
second moments, so one step needs one exchange: an
all-reduce(sum) of the packed fp64 moment buffer between the two halves of the
step (``cesx_moments`` -> all-reduce -> ``cesx_apply``).  The buffer is sent in
two pieces -- first the part that depends on U alone (N, sum u, S_uu: 0.5 MB
at p = 256), then the rest -- so that every rank can start chol(C) on its side
stream while the remaining 74 % of the Gram is still being computed; on GPUs the
first piece is itself reduced on a second stream beside that Gram launch, and the
total payload is that of one all-reduce.  A step is exposed as two halves,
``begin`` (moments, all-reduces, chol(C)) and ``finish`` (K2 with the pseudo-time
of the previous step, K3), so that a driver can enqueue ``begin`` of step i+1
before it reads the result of step i.  Rule-specific extras:
a (1+p+n)-double all-reduce when the centring shift is (re)computed from the
data (first step of a run), and a one-scalar all-reduce(max) for
``eks_update_aldi_constant`` (ces/calibrate.py:519 takes max|drift| over the
whole ensemble).  The two fourth-order data metrics need the exact mean, so a
shard can only sum them AFTER the all-reduce; its two sums ride at the tail of
the NEXT step's moment buffer (``StepResult.lag_*``) and are flushed once at the
end of a run -- no per-step second collective.  Collectives go through
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).

The engine object only needs the split entry points of the C ABI, so the same
driver is exercised on CPU with an oracle-backed stand-in (tests/).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

_METRIC_KEYS = ("self-bias", "self-bias-data", "bias-data", "bias", "t")


def shard_range(J, world, rank):
    """Balanced contiguous column range of rank ``rank`` (first J % world ranks get one extra)."""
    base, extra = divmod(int(J), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedUpdate:
    """Drives one engine (this rank's shard) through sharded steps."""

    def __init__(self, engine, group=None, overlap_comm=None, single_allreduce=None, native_comm=None):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._recentered = False
        self._last_begin = None           # how to redo the `begin` enqueued behind the last `finish` (result(): a re-run step)
        self.native_comm = False          # set below: the all-reduces go through the engine's own RCCL communicator (cesx_allreduce_*)
        # CESX_SINGLE_ALLREDUCE=1 (or single_allreduce=True): the north star's literal form -- ONE all-reduce of the
        # whole moment buffer per step, after the complete Gram, with chol(C) in line behind it (on the critical
        # path).  The default splits the same payload in two (head beside the second Gram launch, then the rest) so
        # that chol(C) runs beside the Gram; which of the two wins on a real xGMI ring is for the 8-GPU run to say
        # (bench.py prints the collective times and the Gram-end -> K3-start gap of a sampled step for both).
        if single_allreduce is None:
            single_allreduce = os.environ.get("CESX_SINGLE_ALLREDUCE", "0") == "1"
        self.single_allreduce = bool(single_allreduce)
        self.n_collectives = 0            # all-reduces issued so far (tests assert the per-step count)
        self.collective_doubles = 0       # ... and their total payload
        self.sample_collectives = False   # record an event pair around every all-reduce of the current step
        self._coll_events = []
        # On GPUs the first all-reduce (the U x U head of the moment buffer) and chol(C) run on a
        # second stream while the main stream goes on with the rest of the Gram: the collective
        # costs no GPU idle time.  CESX_FORCE_COMM_OVERLAP=1 takes that path on one rank too (tests).
        if overlap_comm is None:
            dev = getattr(engine, "device", None)
            overlap_comm = (isinstance(dev, torch.device) and dev.type == "cuda" and
                            (self.world > 1 or os.environ.get("CESX_FORCE_COMM_OVERLAP") == "1"))
        self.overlap_comm = bool(overlap_comm)
        dev = getattr(engine, "device", None)
        self._cs = None
        self._moms, self._mom_idx = None, 0
        self._nuu = None
        # one-rank rehearsal of the multi-GPU path: issue the collectives even though world == 1
        self._force_collectives = dist.is_initialized() and os.environ.get("CESX_FORCE_COLLECTIVES") == "1"
        # The exchange step behind the C ABI (include/cesx.h, cesx_comm_* / cesx_allreduce_*): on GPUs with the "nccl"
        # backend the engine gets an RCCL communicator of its own -- rank 0 draws the id, torch.distributed only carries
        # those 128 bytes to the other ranks -- and every all-reduce of a step is issued by the library on the stream it
        # belongs to.  torch.distributed stays the path of the CPU (gloo) tests and of CESX_NATIVE_COMM=0.
        if native_comm is None:
            native_comm = (os.environ.get("CESX_NATIVE_COMM", "1") != "0" and hasattr(engine, "comm_init")
                           and isinstance(dev, torch.device) and dev.type == "cuda" and dist.is_initialized()
                           and dist.get_backend(group) == "nccl" and (self.world > 1 or self._force_collectives))
        if native_comm:
            self._init_native_comm()

    def _init_native_comm(self):
        eng = self.engine
        if eng.comm_nranks() == 0:
            rank = dist.get_rank(self.group) if dist.is_initialized() else 0
            box = [eng.comm_unique_id() if rank == 0 else None]
            if self.world > 1:
                src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                dist.broadcast_object_list(box, src=src, group=self.group)
            eng.comm_init(self.world, rank, box[0])
        if eng.comm_nranks() != self.world:
            raise RuntimeError("the engine's communicator spans %d ranks, the process group %d" % (eng.comm_nranks(), self.world))
        self.native_comm = True

    def _all_reduce(self, t, op=dist.ReduceOp.SUM, tag="moments", mom=None):
        """``mom``: the whole moment buffer when ``t`` is its head / tail / whole (``tag``) -- the engine's communicator
        then sums that piece itself (cesx_allreduce_head / _tail / _whole)."""
        if self.world > 1 or self._force_collectives:
            self.n_collectives += 1
            self.collective_doubles += int(t.numel())
            timed = self.sample_collectives and t.is_cuda
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(torch.cuda.current_stream(t.device))
            if self.native_comm and t.is_cuda:
                if mom is not None and tag in ("head", "tail", "whole"):
                    self.engine.allreduce(mom, part=tag)
                else:
                    self.engine.allreduce(t, op="max" if op == dist.ReduceOp.MAX else "sum")
            else:
                dist.all_reduce(t, op=op, group=self.group)
            if timed:
                b.record(torch.cuda.current_stream(t.device))
                self._coll_events.append((tag, int(t.numel()), a, b))
        return t

    def read_collective_ms(self):
        """{tag: [doubles, ms]} of the all-reduces sampled since the last call (``sample_collectives``); synchronises."""
        out = {}
        for tag, numel, a, b in self._coll_events:
            b.synchronize()
            cur = out.setdefault(tag, [0, 0.0])
            cur[0] += numel
            cur[1] += a.elapsed_time(b)
        self._coll_events = []
        return out

    def recenter(self, U, G):
        """Centring shift = global ensemble means (same value on every rank)."""
        sums = self._all_reduce(self.engine.colsum(U, G), tag="recenter")
        self.engine.set_shift(sums)
        self._recentered = True

    def _moment_buffer(self):
        """Two persistent moment buffers, used alternately.  The buffer of a step is written on the
        main stream and read / reduced on the side stream; owning the buffers for the lifetime of the
        driver (instead of allocating one per step) means the caching allocator never hands a buffer
        that a stream may still be reading to a new allocation."""
        eng = self.engine
        if self._moms is None:
            if hasattr(eng, "device") and isinstance(eng.device, torch.device):
                self._moms = [torch.zeros(eng.moments_len(), dtype=torch.float64, device=eng.device) for _ in range(2)]
            else:
                self._moms = [torch.zeros(eng.moments_len(), dtype=torch.float64) for _ in range(2)]
        self._mom_idx ^= 1
        return self._moms[self._mom_idx]

    def begin(self, prm, U, G, recenter=False, noise_step=None, forward=None):
        """First half of a step: everything that does not need the pseudo-time of the previous
        step -- the moments, their all-reduce and chol(C).  Only ``prm.update`` is read, so a
        driver may enqueue ``begin`` of step i+1 BEFORE it reads the result of step i: the
        host's read (ces/calibrate.py:387 tests ``t`` every iteration) then overlaps the Gram
        of the next step instead of idling the GPU.  ``noise_step``: Philox step index of the
        update this ``begin`` belongs to -- its noise block (and that of the next step) is then drawn
        ahead on the engine's side stream (cesx_prefetch_noise) instead of inside the update kernel.

        Whether the head all-reduce runs on the side stream is decided ONCE, at construction
        (``overlap_comm``): a collective that may already have been enqueued is never retried,
        and an error raised here (HIP, RCCL) propagates, so that all ranks fail together instead
        of one rank issuing an extra collective.

        ``forward``: ``forward(U, out=G)`` re-evaluates the forward map INTO ``G`` (a pipelined driver that computed
        ``G = forward(U)`` from an ensemble the previous step was still to write passes it): should ``result()`` find
        that the previous step had to be re-run, ``G`` was computed from an unwritten ``U`` -- the redo of this
        ``begin`` then refreshes ``G`` in place first, so the driver's ``finish(prm, U, G)`` reads the right one."""
        eng = self.engine

        def redo():
            if forward is not None:
                forward(U, out=G)
            self.begin(prm, U, G, recenter=False, noise_step=noise_step, forward=forward)
        self._last_begin = redo
        if recenter or not self._recentered:
            self.recenter(U, G)
        if noise_step is not None and hasattr(eng, "prefetch_noise") and not self.single_allreduce:
            eng.prefetch_noise(noise_step)       # (drawn behind chol(C) on the side stream: the single-collective mode has none)
        nuu = self._nuu
        if nuu is None:                          # (constant for the engine's lifetime: one C call, not one per step)
            nuu = self._nuu = eng.moments_uu_len()
        mom = self._moment_buffer()
        if self.single_allreduce:
            # the whole Gram, ONE all-reduce of the whole buffer; cesx_apply then finds no chol(C) in flight and
            # factors C in line (K2 of launch_dense) -- no side stream, no second collective
            eng.moments_uu(U, G, out=mom)             # (= cesx_moments: both Gram launches back to back)
            eng.moments_rest(U, G, mom)
            self._all_reduce(mom, tag="whole", mom=mom)
            self._mom = mom
            return mom
        if self.overlap_comm and hasattr(eng, "moments_uu_handover"):
            # main stream: U x U Gram (the device to itself) -> hand-over -> the rest of the Gram;
            # side stream (high priority), beside the rest of the Gram: all-reduce of the head -> C, L = chol(C)
            # (the U x U launch itself on the side stream was measured slower -- 0.54 against 0.49 ms/step at C2, round 2: the
            #  two Gram launches fight for the CUs, the U x U one finishes late and chol(C) starts late -- and is gone)
            if self._cs is None:        # the engine's own side stream: no third stream to share a hardware queue
                self._cs = eng.side_stream()
            eng.moments_uu_handover(U, G, out=mom)
            with torch.cuda.stream(self._cs):
                self._all_reduce(mom[:nuu], tag="head", mom=mom)  # N, sum(u - s), S_aa: all chol(C) needs
                eng.chol_async(prm, mom)
            eng.moments_rest(U, G, mom)
        elif self.world == 1 and not self._force_collectives and hasattr(eng, "moments_uu_chol"):
            eng.moments_uu_chol(prm, U, G, out=mom)  # no collective between the two: one call, no marker packet
            eng.moments_rest(U, G, mom)
        else:
            eng.moments_uu(U, G, out=mom)
            self._all_reduce(mom[:nuu], tag="head", mom=mom)
            eng.chol_async(prm, mom)                 # C, then L = chol(C) on the side stream ...
            eng.moments_rest(U, G, mom)              # ... beside the rest of the Gram
        if self.world > 1 or self._force_collectives:          # (one device: not even the tensor view is made)
            self._all_reduce(mom[nuu:], tag="tail", mom=mom)      # apply() joins the side stream before K2 reads the head
        self._mom = mom
        return mom

    def lineal_fast_ok(self, model):
        """The linear-map fast path applies: the model is ``ces_amd.utils.lineal`` with its device hook (the engine
        evaluates G = A U + b itself), the engine has ``cesx_moments_rest_lineal``, and the ensemble is on one device
        (CESX_LINEAL_FAST=0 switches it off)."""
        return (getattr(model, "engine_lineal", False) and not getattr(model, "flag_noise", False)
                and hasattr(self.engine, "moments_rest_lineal") and hasattr(self.engine, "moments_uu_chol")
                and not getattr(self.engine, "dense_gamma", False)       # (whitened data: the full Gram)
                and self.world == 1 and not self._force_collectives and not self.single_allreduce
                and os.environ.get("CESX_LINEAL_FAST", "1") != "0")

    def begin_lineal(self, prm, U, forward, noise_step=None, model=None, out=None):
        """``begin`` for a linear forward map the engine evaluates itself: only the U x U Gram
        runs; G = forward(U) is evaluated on the caller's stream BESIDE chol(C), and every G-dependent moment
        follows from the U-only head (cesx_moments_rest_lineal: two small fp64 products instead of the second Gram
        launch and its reduce -- 100 of the 136 blocks at p = n_obs = 256).  The centring shift must be valid (a
        first step goes through ``begin`` with ``recenter``).  Returns (mom, G).  ``forward(U, out=None)``: a redo
        (``result()``: the previous step was re-run) evaluates it INTO the tensor this call returned, which the driver
        already holds."""
        eng = self.engine
        if model is not None and hasattr(model, "ensure_installed"):
            model.ensure_installed(eng)              # the moment kernels below read the INSTALLED map: this model's, as of now
        if noise_step is not None:
            eng.prefetch_noise(noise_step)
        mom = self._moment_buffer()
        eng.moments_uu_chol(prm, U, U, out=mom)      # (the G argument is not read by the U x U launch)
        # the moment kernels FIRST (they need A and the head, not G): they are small, so the side stream's centring and
        # chol(C) find their CUs while they run, and the forward GEMM behind fills what is left.  With the GEMM first --
        # 512 workgroups, two on every CU -- the factorisation (8 waves x 256 registers: a CU to itself) started when the
        # GEMM had drained: 0.477 against 0.457 ms/step at C2 (NOTEBOOK.md section 6)
        eng.moments_rest_lineal(mom)
        G = forward(U) if out is None else forward(U, out=out)
        self._mom = mom
        self._last_begin = lambda: self.begin_lineal(prm, U, forward, noise_step=noise_step, model=model, out=G)
        return mom, G

    def finish(self, prm, U, G, xi=None, out=None):
        """Second half: K2 with this step's parameters (t_last, time-step rule) and K3."""
        self._last_begin = None                 # (a `begin` enqueued from here on belongs to the NEXT step: only that one is ever redone)
        eng, mom = self.engine, self._mom
        if prm.update == 2:                     # aldi_constant: max|drift| over all shards
            out = eng.empty(eng.p) if out is None else out
            absmax = self._all_reduce(eng.apply_drift(prm, mom, U, G, out), op=dist.ReduceOp.MAX, tag="absmax")
            return eng.apply_finish(prm, absmax, U, xi, out)
        return eng.apply(prm, mom, U, G, xi=xi, out=out)

    def step(self, prm, U, G, xi=None, out=None, recenter=False):
        """moments -> all-reduce -> apply on this rank's shard.  Returns U_next."""
        self.begin(prm, U, G, recenter=recenter, noise_step=int(prm.step_index) if xi is None else None)
        return self.finish(prm, U, G, xi=xi, out=out)

    def result(self):
        """Result of the last step.  On more than one rank ``bias_data`` /
        ``self_bias_data`` are this shard's share; ``lag_*`` are the complete values of
        the previous step."""
        try:
            return self.engine.result()
        except Exception as err:
            # cesx_result re-ran a step whose polled join of the side stream had run out (a profiler that serialises the
            # streams does that to it) and says that the `begin` enqueued behind it read an ensemble that had not been
            # written: redo that `begin` (the step's own result is valid and attached), carry on with the event join
            redo, res = self._last_begin, getattr(err, "result", None)
            if res is None or redo is None or getattr(err, "code", None) != 4:
                raise
            self.redone_begins = getattr(self, "redone_begins", 0) + 1
            self._last_begin = None
            redo()
            return res

    def flush_data_metrics(self, res):
        """Complete data metrics of the LAST step (one tiny all-reduce at the end of a run)."""
        t = torch.tensor([res.bias_data, res.self_bias_data], dtype=torch.float64)
        if self.world > 1:
            dev = getattr(self.engine, "device", None)
            t = t.to(dev) if dev is not None and dist.get_backend(self.group) == "nccl" else t
            dist.all_reduce(t, group=self.group)
        return float(t[0]), float(t[1])


class ShardedSampler:
    """``sampling.run`` (ces/calibrate.py:270-416) for an ensemble sharded over ranks.

    Every rank holds the columns ``shard_range(J, world, rank)`` of the ensemble
    on its device for the whole run; the forward map is evaluated per shard
    (``model.forward_device`` when the model offers the hook, otherwise the
    reference's host loop over the shard's particles).  All ranks compute the
    same hk / t / metrics (K2 runs redundantly on identical all-reduced moments),
    so every rank takes the same ``t_tol`` decision without a broadcast.
    """

    def __init__(self, engine, p, n_obs, J, group=None, single_allreduce=None):
        self.engine, self.p, self.n_obs, self.J = engine, p, n_obs, J
        self.sh = ShardedUpdate(engine, group, single_allreduce=single_allreduce)
        self.T = 30
        self.metrics = {k: [] for k in _METRIC_KEYS}
        self.radspec = []
        self._steps_done = 0           # Philox step counter: a resumed run() draws fresh noise

    def _forward(self, model, U, out=None):
        if hasattr(model, "forward_device"):
            if out is None:
                return model.forward_device(self.engine, U)
            from .utils import hook_takes_out
            if hook_takes_out(model.forward_device):
                return model.forward_device(self.engine, U, out=out)
            out.copy_(model.forward_device(self.engine, U))         # a hook without ``out=``
            return out
        Uh = U.cpu().numpy().astype(np.float64)
        Gh = np.stack([np.asarray(model(u)) for u in Uh.T], axis=1)
        G = self.engine.to_device(Gh[: self.n_obs])
        if out is not None:
            out.copy_(G)
            return out
        return G

    def run(self, y_obs, U_shard, model, Gamma, mu, sigma, ustar, update="aldi", xis=None, **kwargs):
        """Returns the final shard (a device tensor of the engine)."""
        from .engine import step_params
        eng, m = self.engine, self.metrics
        eng.set_problem(y_obs, Gamma, mu, sigma, ustar)
        U = eng.to_device(U_shard)
        res = None
        # With a device-side forward map nothing in an iteration needs the host until the t_tol
        # test, so the first half of step i+1 (forward map, moments, all-reduce, chol) is enqueued
        # before the result of step i is read; if the run stops there, that work is discarded.
        pipelined = hasattr(model, "forward_device")
        fast = pipelined and self.sh.lineal_fast_ok(model)      # G's moments from U's (linear map on the device)
        prm0 = step_params(update=update, T=self.T)
        G = self._forward(model, U)
        draw = xis is None                     # on-device noise: drawn ahead of each update
        if pipelined:
            self.sh.begin(prm0, U, G, recenter=True, noise_step=self._steps_done if draw else None)
        for i in range(self.T):
            t = m["t"]
            prm = step_params(update=update, time_step=kwargs.get("time_step"), first_step=(i == 0 and not t),
                              t_len=len(t), t_last=t[-1] if t else 0.0, delta_t=kwargs.get("delta_t"),
                              spinup=kwargs.get("spinup", 4.0), switch=kwargs.get("switch", 1.0),
                              step_index=self._steps_done, T=self.T)
            self._steps_done += 1
            xi = None if xis is None else eng.to_device(xis[i])
            if pipelined:
                U = self.sh.finish(prm, U, G, xi=xi)
                if i + 1 < self.T:
                    fwd = lambda u, out=None: self._forward(model, u, out=out)       # noqa: E731
                    if fast:
                        _, G = self.sh.begin_lineal(prm0, U, fwd, model=model,
                                                    noise_step=self._steps_done if draw else None)
                    else:
                        G = self._forward(model, U)
                        self.sh.begin(prm0, U, G, noise_step=self._steps_done if draw else None, forward=fwd)
            else:
                U = self.sh.step(prm, U, G, xi=xi, recenter=(i == 0))
            res = self.sh.result()
            m["self-bias"].append(res.self_bias)
            m["bias"].append(res.bias)
            m["t"].append(res.t_new)
            if kwargs.get("time_step") == "spectral" and update != "aldi_constant":
                self.radspec.append(res.radspec)
            if self.sh.world == 1:
                m["bias-data"].append(res.bias_data)
                m["self-bias-data"].append(res.self_bias_data)
            elif i > 0:                          # the previous step's complete values arrive now
                m["bias-data"].append(res.lag_bias_data)
                m["self-bias-data"].append(res.lag_self_bias_data)
            if m["t"][-1] > kwargs.get("t_tol", 2.0):
                break
            if not pipelined and i + 1 < self.T:
                G = self._forward(model, U)
        if self.sh.world > 1 and res is not None:
            bd, sbd = self.sh.flush_data_metrics(res)
            m["bias-data"].append(bd)
            m["self-bias-data"].append(sbd)
        self.Ustar = U
        return U
