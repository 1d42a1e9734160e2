"""Multi-rank path on CPU: world_size-2 gloo processes drive ces_amd.dist with an
oracle-backed stand-in for the device shard (oracle/fake_engine.py) and must
reproduce the single-process oracle (one all-reduce(sum) of the
packed moments per step; max-reduction for aldi_constant; lagged data metrics)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(p, n, J, T, seed=3):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    B = rng.standard_normal((n, n))
    Gamma = 0.05 * (B @ B.T / n + np.eye(n))
    sigma = 4.0 * np.eye(p)
    mu = 0.1 * rng.standard_normal((p, 1))
    y = (A @ ustar).ravel() + 0.2 * rng.standard_normal(n)
    U0 = ustar + rng.standard_normal((p, J))
    xis = rng.standard_normal((T, p, J))
    return dict(A=A, ustar=ustar, Gamma=Gamma, sigma=sigma, mu=mu, y=y, U0=U0, xis=xis)


def _worker(rank, world, port, update, kwargs, q, device_hook=False, dims=(5, 4, 37, 6), overlap=False, single=False):
    sys.path.insert(0, ROOT)
    from ces_amd.dist import ShardedSampler, shard_range
    from ces_amd.utils import lineal
    from oracle.fake_engine import FakeEngine
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    p, n, J, T = dims
    d = _problem(p, n, J, T)
    lo, hi = shard_range(J, world, rank)
    eng = FakeEngine(p, n, hi - lo, J_global=J, j_offset=lo)
    smp = ShardedSampler(eng, p, n, J, single_allreduce=single)
    smp.T = T
    if overlap:                              # the N > 1 GPU branch of ShardedUpdate.begin (hand-over + side-stream collective)
        smp.sh.overlap_comm = True
    class HostLineal:                        # the stand-in has no device hook: host loop per particle
        type, n_obs, model_name = "map", n, "lineal"

        def __call__(self, theta):
            return lineal(d["A"])(theta)
    class HookLineal(HostLineal):            # with a shard-wide hook ShardedSampler pipelines the loop:
        def forward_device(self, engine, U):  # begin(i+1) is enqueued before result(i) is read
            return engine.to_device(d["A"] @ U.numpy())
    model = HookLineal() if device_hook else HostLineal()
    try:
        U = smp.run(d["y"], d["U0"][:, lo:hi], model, d["Gamma"], d["mu"], d["sigma"], d["ustar"],
                    update=update, xis=d["xis"][:, :, lo:hi], t_tol=1e9, **kwargs)
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, U.numpy()))
        if overlap:
            assert eng.calls.count("uu_handover") >= T, eng.calls
        # collectives per step: the recentring of the first step apart, ONE all-reduce of the whole buffer in the
        # north star's literal mode, two (head + tail, the same payload) by default; aldi_constant adds its max
        steps = len(smp.metrics["t"])
        per_step = (1 if single else 2) + (1 if update == "aldi_constant" else 0)
        assert smp.sh.n_collectives == 1 + per_step * steps, (smp.sh.n_collectives, steps)
        assert smp.sh.collective_doubles == (1 + p + n) + steps * (eng.moments_len() + (1 if update == "aldi_constant" else 0))
        if rank == 0:
            full = np.concatenate([g[1] for g in sorted(gathered, key=lambda g: g[0])], axis=1)
            q.put((full, {k: list(v) for k, v in smp.metrics.items()}, list(smp.radspec)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("update,kwargs,hook", [("aldi", {}, False), ("eks", {"time_step": "constant", "delta_t": 0.01}, False),
                                                ("aldi_constant", {"switch": 0.5}, False),
                                                ("aldi", {"time_step": "spectral"}, False),
                                                ("aldi", {}, True), ("aldi_constant", {"switch": 0.5}, True)])
def test_two_ranks_match_single_process_oracle(update, kwargs, hook):
    from oracle import ces_numpy as oc
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, update, kwargs, q, hook)) for r in range(world)]
    for pr in procs:
        pr.start()
    full, metrics, radspec = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p, n, J, T = 5, 4, 37, 6
    d = _problem(p, n, J, T)
    st = oc.OracleState(p, n, J, d["mu"], d["sigma"], d["ustar"], T=T)
    Uall, _ = oc.run_chain(st, d["y"], d["U0"], lambda U: oc.lineal_forward(d["A"], U), d["Gamma"], d["xis"],
                           update=update, step=oc.factored_step, t_tol=1e9, **kwargs)
    assert np.allclose(full, Uall[-1], rtol=1e-9, atol=1e-12)
    for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t"):
        assert len(metrics[k]) == T
        assert np.allclose(metrics[k], st.metrics[k], rtol=1e-9), k
    if kwargs.get("time_step") == "spectral":
        assert np.allclose(radspec, st.radspec, rtol=1e-8)


def _check_against_oracle(full, metrics, radspec, dims, update, kwargs):
    from oracle import ces_numpy as oc
    p, n, J, T = dims
    d = _problem(p, n, J, T)
    st = oc.OracleState(p, n, J, d["mu"], d["sigma"], d["ustar"], T=T)
    Uall, _ = oc.run_chain(st, d["y"], d["U0"], lambda U: oc.lineal_forward(d["A"], U), d["Gamma"], d["xis"],
                           update=update, step=oc.factored_step, t_tol=1e9, **kwargs)
    assert np.allclose(full, Uall[-1], rtol=1e-9, atol=1e-12)
    for k in ("self-bias", "self-bias-data", "bias-data", "bias", "t"):
        assert len(metrics[k]) == T
        assert np.allclose(metrics[k], st.metrics[k], rtol=1e-9), k


@pytest.mark.parametrize("update,kwargs,hook", [("aldi", {}, True), ("aldi", {}, False), ("aldi_constant", {"switch": 0.5}, True)])
def test_two_ranks_stream_overlap_branch(update, kwargs, hook):
    """The branch of ShardedUpdate.begin that N > 1 GPU runs take (U x U moments with the hand-over on the caller's
    stream, all-reduce of the head + chol(C) under the side-stream context, the rest of the moments, the second
    all-reduce) driven by two gloo ranks on the CPU: the same collectives in the same order on every rank, the same
    numbers as the single-process oracle."""
    dims = (5, 4, 37, 6)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, update, kwargs, q, hook, dims, True)) for r in range(world)]
    for pr in procs:
        pr.start()
    full, metrics, radspec = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    _check_against_oracle(full, metrics, radspec, dims, update, kwargs)


@pytest.mark.parametrize("update,kwargs,hook", [("aldi_constant", {"switch": 0.5}, True), ("aldi", {}, False)])
def test_eight_ranks_ragged_shards(update, kwargs, hook):
    """World 8 (the node the driver scales to) with J = 43 not divisible by 8: shards of 6 and 5
    particles, the aldi_constant max-reduction and the lagged data metrics over 8 ranks."""
    world, port, dims = 8, _free_port(), (5, 4, 43, 4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, update, kwargs, q, hook, dims)) for r in range(world)]
    for pr in procs:
        pr.start()
    full, metrics, radspec = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    _check_against_oracle(full, metrics, radspec, dims, update, kwargs)


@pytest.mark.parametrize("update,kwargs,hook", [("aldi", {}, True), ("eks", {}, False), ("aldi_constant", {"switch": 0.5}, True)])
def test_single_allreduce_mode_matches_the_split_mode(update, kwargs, hook):
    """CESX_SINGLE_ALLREDUCE (the north star's literal form: one all-reduce of the whole moment buffer per step,
    chol(C) in line) and the default two-piece form give the same trajectory and metrics on two gloo ranks -- and
    both equal the single-process oracle; the per-step collective count (1 vs 2) is asserted in the worker."""
    dims = (5, 4, 37, 6)
    res = []
    for single in (True, False):
        world, port = 2, _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, update, kwargs, q, hook, dims, False, single)) for r in range(world)]
        for pr in procs:
            pr.start()
        res.append(q.get(timeout=120))
        for pr in procs:
            pr.join(timeout=60)
            assert pr.exitcode == 0
    (full1, m1, _), (full2, m2, _) = res
    assert np.array_equal(full1, full2)
    for k in m1:
        assert m1[k] == m2[k], k
    _check_against_oracle(full1, m1, None, dims, update, kwargs)


def test_resumed_run_draws_fresh_noise():
    """A second ShardedSampler.run on the same sampler continues the Philox step counter (the
    single-device class keeps ``_step_counter``): it must not replay the first run's xi blocks."""
    sys.path.insert(0, ROOT)
    from ces_amd.dist import ShardedSampler
    from ces_amd.utils import lineal
    from oracle.fake_engine import FakeEngine
    p, n, J, T = 5, 4, 37, 3
    d = _problem(p, n, J, T)
    eng = FakeEngine(p, n, J)
    smp = ShardedSampler(eng, p, n, J)
    smp.T = T
    class model:                         # host loop per particle (the stand-in has no device hook)
        type, n_obs, model_name = "map", n, "lineal"

        def __call__(self, theta):
            return lineal(d["A"])(theta)
    model = model()
    U1 = smp.run(d["y"], d["U0"], model, d["Gamma"], d["mu"], d["sigma"], d["ustar"], t_tol=1e9)
    U2 = smp.run(d["y"], U1, model, d["Gamma"], d["mu"], d["sigma"], d["ustar"], t_tol=1e9)
    assert eng.drawn_steps == list(range(2 * T))
    assert len(smp.metrics["t"]) == 2 * T and smp.metrics["t"][T] > smp.metrics["t"][T - 1]
    assert not np.allclose(U1.numpy(), U2.numpy())


def test_shard_range_covers_everything():
    from ces_amd.dist import shard_range
    for J, w in ((37, 2), (65536, 8), (10, 3), (7, 8)):
        cuts = [shard_range(J, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == J
        assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
        assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


def test_bench_rank_control_flow_on_two_gloo_ranks():
    """bench.py as the driver launches it for N > 1 (``python -m torch.distributed.run --nproc-per-node 2 ... bench.py
    --gpus 2``), with the platform swapped for the CPU stand-in (tests/bench_rehearsal.py: gloo, the numpy engine): both
    ranks run the same number of pre-warm windows, pass the barriers, reduce the elapsed time, rank 0 prints ONE JSON
    line with n_gpus == 2, and both processes end with status 0 -- no hang at the first contact of an 8-GPU run."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CESX_BENCH_PREWARM_S="0", CESX_BENCH_NOPROF="1", OMP_NUM_THREADS="2", CESX_NATIVE_COMM="0",
               BENCH_REHEARSAL_ARGS="--p 8 --n 6 --J 64 --dtype float64 --no-cpu-baseline --no-extras")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "bench_rehearsal.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1"]
    for pin in (None, "0"):
        # (no pin: both ranks time both collective modes behind the pre-warm and keep the same, faster one -- round 6)
        env_run = dict(env) if pin is None else dict(env, CESX_SINGLE_ALLREDUCE=pin)
        env_run.pop("CESX_SINGLE_ALLREDUCE", None) if pin is None else None
        res = subprocess.run(cmd, env=env_run, cwd=root, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0
        ss = rec["sampled_step"]
        assert rec["config"]["J_global"] == 128
        if pin is None:
            mc = ss["mode_choice"]
            assert set(mc["ms_per_step"]) == {"head_tail", "single"} and mc["picked"] == min(mc["ms_per_step"], key=mc["ms_per_step"].get)
            assert ss["collectives_per_step"] == (1 if mc["picked"] == "single" else 2)
            assert "mode measured and chosen" in rec["config"]["parallelism"]
        else:
            assert ss["mode_choice"] is None and ss["collectives_per_step"] == 2 and "pinned" in rec["config"]["parallelism"]
        assert rec["prewarm_steps"] == 512 and rec["scaling"] == "weak"
