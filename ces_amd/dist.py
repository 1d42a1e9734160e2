"""Particle-sharded data parallelism for the ensemble update.

One process per GPU.  Rank r owns the column range ``shard_range(J, N, r)`` of
U, G and U_next.  Particles are exchangeable and couple only through first and
second moments, so one step needs exactly ONE collective: an
all-reduce(sum) of the packed fp64 moment buffer between the two halves of the
step (``cesx_moments`` -> all-reduce -> ``cesx_apply``).  Two rule-specific
extras: a (1+p+n)-double all-reduce when the centring shift is (re)computed
from the data (first step of a run), and a one-scalar all-reduce(max) for
``eks_update_aldi_constant`` (ces/calibrate.py:519 takes max|drift| over the
whole ensemble).  The collectives go through ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The engine object only needs the split entry points of the C ABI, so the same
driver is exercised on CPU with an oracle-backed stand-in (tests/).
"""
import torch
import torch.distributed as dist


def shard_range(J, world, rank):
    """Balanced contiguous column range of rank ``rank`` (first J % world ranks get one extra)."""
    base, extra = divmod(int(J), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedUpdate:
    """Drives one engine (this rank's shard) through sharded steps."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._recentered = False

    def _all_reduce(self, t, op=dist.ReduceOp.SUM):
        if self.world > 1:
            dist.all_reduce(t, op=op, group=self.group)
        return t

    def recenter(self, U, G):
        """Centring shift = global ensemble means (same value on every rank)."""
        sums = self._all_reduce(self.engine.colsum(U, G))
        self.engine.set_shift(sums)
        self._recentered = True

    def step(self, prm, U, G, xi=None, out=None, recenter=False):
        """moments -> all-reduce -> apply on this rank's shard.  Returns U_next."""
        eng = self.engine
        if recenter or not self._recentered:
            self.recenter(U, G)
        mom = self._all_reduce(eng.moments(U, G))
        if prm.update == 2:                     # aldi_constant: max|drift| over all shards
            out = eng.empty(eng.p) if out is None else out
            absmax = self._all_reduce(eng.apply_drift(prm, mom, U, G, out), op=dist.ReduceOp.MAX)
            return eng.apply_finish(prm, absmax, U, xi, out)
        return eng.apply(prm, mom, U, G, xi=xi, out=out)
