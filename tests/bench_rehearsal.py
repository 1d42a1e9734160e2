"""Run ``bench.py``'s main() on the CPU: gloo instead of RCCL, the numpy stand-in engine of the CPU tests instead of
libcesx.so.  Started by tests/test_dist_gloo.py under ``torch.distributed.run`` with two processes: what is exercised is
bench.py's RANK CONTROL FLOW -- communicator set-up with stdout parked, the pre-warm decision every rank takes alike
(an all-reduce(MIN) per window), the barriers on both sides of the timed region, the all-reduce(MAX) of the elapsed
time, the one JSON line of rank 0 and the tear-down of the other ranks -- not a number.  TEST INFRASTRUCTURE."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle.fake_engine import FakeEngine  # noqa: E402


class _Engine(FakeEngine):
    """The stand-in shard with the few extra entry points bench.py's main() calls (profiling: no-ops)."""
    torch_dtype = torch.float64

    def __init__(self, p, n_obs, J, dtype="float64", device=0, J_global=None, j_offset=0, seed=1234):
        super().__init__(p, n_obs, J, J_global=J_global, j_offset=j_offset, seed=seed)
        self.device = torch.device("cpu")

    def forward_lineal(self, A, U):
        return torch.as_tensor(np.asarray(A, dtype=np.float64)) @ U

    def prefetch_noise(self, step):
        pass

    def profile_enable(self, mode):
        pass

    def profile_read(self, which):
        return 1.0, 1

    def profile_gap(self):
        return 0.01

    def profile_clock(self):
        return 2.0

    def calibrate_mfma(self, ms):
        return 100.0, 2.0

    def update_form(self):
        return 0


class _Module:
    Engine = _Engine

    @staticmethod
    def step_params(**kw):
        from ces_amd import engine          # (the ctypes struct alone: libcesx.so is not loaded for it)
        return engine.step_params(**kw)


class CpuPlatform(bench.Platform):
    backend = "gloo"

    def device(self, local):
        return torch.device("cpu")

    def sync(self):
        pass

    def modules(self):
        from ces_amd.dist import ShardedUpdate
        return (lambda: None), _Module, ShardedUpdate

    def generator(self, dev):
        return torch.Generator()


if __name__ == "__main__":
    # (sizes come through the environment: torch.distributed.run's own parser trips over "--n" behind the script name)
    sys.argv += os.environ.get("BENCH_REHEARSAL_ARGS", "").split()
    bench.PLATFORM = CpuPlatform()
    bench.main()
