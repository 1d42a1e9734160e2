"""RCCL inside ``pytest -m gpu``: the N > 1 code path of ces_amd.dist (head all-reduce + chol(C) on the
engine's side stream beside the second Gram launch, tail all-reduce on the main stream, max-reduction for
aldi_constant) with the collectives really issued through torch.distributed's "nccl" backend (= RCCL) on a
ONE-rank communicator ("RCCL path exercised with nranks = 1"), and -- on a box with two
devices -- a two-rank run against the single-rank result.  This file sorts last on purpose: RCCL creates its
own streams when the communicator comes up."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(p, n, J, seed=31):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    U0 = ustar + 0.5 * rng.standard_normal((p, J))
    return dict(A=A, ustar=ustar, Gamma=0.01 * np.eye(n), sigma=100.0 * np.eye(p), mu=np.zeros((p, 1)), y=y, U0=U0)


@pytest.fixture(scope="module")
def rccl_one_rank():
    import torch
    import torch.distributed as dist
    from ces_amd import build
    build.build_lib()
    assert torch.cuda.is_available()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)                                   # the communicator exists and sees one rank
    assert float(t.sum()) == 4.0 and dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    yield dist
    dist.destroy_process_group()


def _chain(engine, d, update, p, n, J, steps, monkeypatch, collectives, counter, single=False, native=False, stats=None,
           time_step=None, forms=None):
    import torch.distributed as dist
    from ces_amd.dist import ShardedUpdate
    if collectives:
        monkeypatch.setenv("CESX_FORCE_COLLECTIVES", "1")
        monkeypatch.setenv("CESX_FORCE_COMM_OVERLAP", "1")
    else:
        monkeypatch.delenv("CESX_FORCE_COLLECTIVES", raising=False)
        monkeypatch.delenv("CESX_FORCE_COMM_OVERLAP", raising=False)
    monkeypatch.setenv("CESX_NATIVE_COMM", "1" if native else "0")
    real = dist.all_reduce

    def counted(t, *a, **k):
        counter.append(int(t.numel()))
        return real(t, *a, **k)
    monkeypatch.setattr(dist, "all_reduce", counted)
    eng = engine.Engine(p, n, J, dtype="float32", seed=9)
    eng.set_problem(d["y"], d["Gamma"], d["mu"], d["sigma"], d["ustar"])
    sh = ShardedUpdate(eng, single_allreduce=single)
    assert sh.overlap_comm == collectives
    assert sh.native_comm == (native and collectives)
    U = eng.to_device(d["U0"])
    t_last, chain = 0.0, []
    for i in range(steps):
        G = eng.forward_lineal(d["A"], U)
        prm = engine.step_params(update=update, time_step=time_step, first_step=(i == 0), t_len=min(i, 1), t_last=t_last,
                                 step_index=i)
        U = sh.step(prm, U, G, xi=None, recenter=(i == 0))
        res = sh.result()
        if forms is not None:
            forms.append(eng.update_form())
        t_last = res.t_new
        chain.append((res.hk, res.t_new, res.bias_data, res.self_bias_data, res.lag_bias_data))
    monkeypatch.setattr(dist, "all_reduce", real)
    if stats is not None:
        stats.append((sh.n_collectives, sh.collective_doubles) + (eng.comm_stats() if sh.native_comm else (0, 0)))
    return U.cpu().numpy(), chain


@pytest.mark.parametrize("update", ["aldi", "aldi_constant"])
def test_one_rank_rccl_path_is_bit_identical(rccl_one_rank, monkeypatch, update):
    from ces_amd import engine
    p, n, J, steps = 128, 96, 8192, 4
    d = _problem(p, n, J)
    calls_plain, calls_rccl = [], []
    ref = _chain(engine, d, update, p, n, J, steps, monkeypatch, False, calls_plain)
    got = _chain(engine, d, update, p, n, J, steps, monkeypatch, True, calls_rccl)
    assert calls_plain == []                               # world == 1: no collective unless forced
    nuu, nall = 1 + p + p * p, 1 + p + n + p * p + p * n + n * n + 2
    per_step = [nuu, nall - nuu] + ([1] if update == "aldi_constant" else [])
    assert calls_rccl == [1 + p + n] + per_step * steps    # centring shift once, then head + tail (+ max) per step
    assert np.array_equal(ref[0], got[0])
    assert ref[1] == got[1]


@pytest.mark.parametrize("update,time_step", [("aldi", None), ("eks", None), ("aldi", "constant"), ("aldi", "spectral"),
                                              ("eks", "spectral"), ("aldi", "mix")])
def test_one_rank_rccl_path_through_every_rule(rccl_one_rank, monkeypatch, update, time_step):
    """The sharded sequence (hand-over, head all-reduce + chol(C) on the side stream through the side communicator, tail
    all-reduce on the caller's stream through the main one, event join) for the rules whose K2 is NOT the default ALDI tail:
    the SPD inverses of `eks` and of the recomputed gain, lambda_max by repeated squaring -- and, at p = 256 with the
    default step, K3 through the Cholesky factor (update form 2).  Bit-identical to the same chain without collectives."""
    from ces_amd import engine
    p, n, J, steps = 256, 96, 4096, 4
    d = _problem(p, n, J)
    st, forms_a, forms_b = [], [], []
    ref = _chain(engine, d, update, p, n, J, steps, monkeypatch, False, [], time_step=time_step, forms=forms_a)
    got = _chain(engine, d, update, p, n, J, steps, monkeypatch, True, [], native=True, stats=st, time_step=time_step, forms=forms_b)
    nuu, nall = 1 + p + p * p, 1 + p + n + p * p + p * n + n * n + 2
    assert st[0][2] == 1 + 2 * steps and st[0][3] == (1 + p + n) + nall * steps
    assert forms_a == forms_b == [2 if (update, time_step) == ("aldi", None) else 0] * steps
    assert np.array_equal(ref[0], got[0])
    assert ref[1] == got[1]


@pytest.mark.parametrize("update", ["aldi", "aldi_constant"])
@pytest.mark.parametrize("single", [False, True])
def test_one_rank_collectives_behind_the_c_abi(rccl_one_rank, monkeypatch, update, single):
    """The exchange step behind the C ABI (cesx_comm_init / cesx_allreduce_head / _tail / _whole / _sum / _max,
    include/cesx.h): the engine's own RCCL communicator -- one rank here, torch.distributed only carries the id --
    issues every all-reduce of a step on the stream it belongs to.  Bit-identical to the torch.distributed path and to
    the run without collectives; count and payload of the all-reduces asserted on the library's own counters (both
    modes: head + tail, and the north star's single all-reduce of the whole buffer)."""
    from ces_amd import engine
    p, n, J, steps = 128, 96, 8192, 4
    d = _problem(p, n, J)
    st_plain, st_torch, st_native, torch_calls = [], [], [], []
    ref = _chain(engine, d, update, p, n, J, steps, monkeypatch, False, [], stats=st_plain)
    viat = _chain(engine, d, update, p, n, J, steps, monkeypatch, True, torch_calls, single=single, stats=st_torch)
    got = _chain(engine, d, update, p, n, J, steps, monkeypatch, True, [], single=single, native=True, stats=st_native)
    nuu, nall = 1 + p + p * p, 1 + p + n + p * p + p * n + n * n + 2
    per_step = ([nall] if single else [nuu, nall - nuu]) + ([1] if update == "aldi_constant" else [])
    want_calls, want_doubles = 1 + len(per_step) * steps, (1 + p + n) + sum(per_step) * steps
    assert st_plain[0] == (0, 0, 0, 0)
    assert st_torch[0] == (want_calls, want_doubles, 0, 0) and torch_calls == [1 + p + n] + per_step * steps
    assert st_native[0] == (want_calls, want_doubles, want_calls, want_doubles)
    assert np.array_equal(ref[0], got[0]) and np.array_equal(viat[0], got[0])
    assert ref[1] == got[1] and viat[1] == got[1]


def test_comm_entry_points_check_their_arguments(rccl_one_rank):
    """cesx_allreduce_* without a communicator -> CESX_ESTATE; a second cesx_comm_init on a handle -> CESX_ESTATE;
    cesx_comm_destroy twice is harmless."""
    import torch
    from ces_amd import engine
    eng = engine.Engine(16, 8, 256, dtype="float32")
    t = torch.zeros(eng.moments_len(), dtype=torch.float64, device=eng.device)
    with pytest.raises(engine.CesxError, match="cesx_comm_init has not been called"):
        eng.allreduce(t, part="head")
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and eng.comm_nranks() == 0
    eng.comm_init(1, 0, uid)
    assert eng.comm_nranks() == 1
    assert eng.comm_count() == 2          # one communicator per stream (main + side): round 6
    with pytest.raises(engine.CesxError, match="already has a communicator"):
        eng.comm_init(1, 0, uid)
    t[:] = 3.0
    eng.allreduce(t, part="whole")
    eng.allreduce(t[:5].contiguous(), op="max")
    torch.cuda.synchronize()
    assert float(t.min()) == 3.0 and eng.comm_stats() == (2, eng.moments_len() + 5)
    eng.comm_destroy()
    eng.comm_destroy()
    assert eng.comm_nranks() == 0 and eng.comm_count() == 0


@pytest.mark.parametrize("update", ["aldi", "aldi_constant"])
def test_one_rank_rccl_single_allreduce_mode(rccl_one_rank, monkeypatch, update):
    """CESX_SINGLE_ALLREDUCE, the north star's literal form: ONE all-reduce (the whole moment buffer) per step
    through RCCL, chol(C) in line.  Same numbers as the default two-piece form to rounding of the in-line
    Cholesky's position in the stream (it is the same kernel on the same matrix: bit-identical)."""
    from ces_amd import engine
    p, n, J, steps = 128, 96, 8192, 4
    d = _problem(p, n, J)
    calls_split, calls_single = [], []
    ref = _chain(engine, d, update, p, n, J, steps, monkeypatch, True, calls_split)
    got = _chain(engine, d, update, p, n, J, steps, monkeypatch, True, calls_single, single=True)
    nall = 1 + p + n + p * p + p * n + n * n + 2
    per_step = [nall] + ([1] if update == "aldi_constant" else [])
    assert calls_single == [1 + p + n] + per_step * steps       # centring shift once, then ONE all-reduce (+ max) per step
    assert sum(calls_single) == sum(calls_split)                # the same payload as head + tail
    assert np.array_equal(ref[0], got[0])
    assert ref[1] == got[1]


def test_one_rank_rccl_sharded_sampler(rccl_one_rank, monkeypatch):
    """ShardedSampler.run (pipelined loop, device forward hook) with RCCL collectives == without."""
    import torch
    from ces_amd import engine
    from ces_amd.dist import ShardedSampler
    from ces_amd.utils import lineal
    p, n, J, T = 64, 50, 4096, 5
    d = _problem(p, n, J, seed=5)
    outs = []
    monkeypatch.setenv("CESX_LINEAL_FAST", "0")        # (bit-identity of the collective path: both runs take the full Gram)
    for coll in (False, True):
        if coll:
            monkeypatch.setenv("CESX_FORCE_COLLECTIVES", "1")
            monkeypatch.setenv("CESX_FORCE_COMM_OVERLAP", "1")
        eng = engine.Engine(p, n, J, dtype="float64", seed=3)
        smp = ShardedSampler(eng, p, n, J)
        smp.T = T
        U = smp.run(d["y"], d["U0"], lineal(d["A"]), d["Gamma"], d["mu"], d["sigma"], d["ustar"], t_tol=1e9)
        torch.cuda.synchronize()
        outs.append((U.cpu().numpy(), dict(smp.metrics)))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


def _two_rank_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            device_id=torch.device("cuda", rank))
    from ces_amd import engine
    from ces_amd.dist import ShardedSampler, shard_range
    from ces_amd.utils import lineal
    p, n, J, T = 64, 50, 4097, 4
    d = _problem(p, n, J, seed=5)
    lo, hi = shard_range(J, world, rank)
    eng = engine.Engine(p, n, hi - lo, dtype="float64", device=rank, J_global=J, j_offset=lo, seed=3)
    smp = ShardedSampler(eng, p, n, J)
    smp.T = T
    U = smp.run(d["y"], d["U0"][:, lo:hi], lineal(d["A"]), d["Gamma"], d["mu"], d["sigma"], d["ustar"], t_tol=1e9)
    assert eng.comm_nranks() == world and eng.comm_stats()[0] > 0      # the collectives went through the library's own communicator
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, U.cpu().numpy()))
    if rank == 0:
        q.put((np.concatenate([g[1] for g in sorted(gathered, key=lambda g: g[0])], axis=1), dict(smp.metrics)))
    dist.destroy_process_group()


def test_two_rank_rccl_matches_single_rank():
    """Two GPUs, two ranks over RCCL vs one rank holding the whole ensemble (fp64, on-device Philox noise
    keyed by the global particle index): same trajectory.  Needs two devices -- skipped on a one-GPU box."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    from ces_amd import engine
    from ces_amd.dist import ShardedSampler
    from ces_amd.utils import lineal
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    full, metrics = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p, n, J, T = 64, 50, 4097, 4
    d = _problem(p, n, J, seed=5)
    eng = engine.Engine(p, n, J, dtype="float64", seed=3)
    smp = ShardedSampler(eng, p, n, J)
    smp.T = T
    U = smp.run(d["y"], d["U0"], lineal(d["A"]), d["Gamma"], d["mu"], d["sigma"], d["ustar"], t_tol=1e9)
    assert np.max(np.abs(U.cpu().numpy() - full)) / np.max(np.abs(full)) < 1e-9
    for k in ("t", "bias", "self-bias", "bias-data", "self-bias-data"):
        assert np.allclose(metrics[k], smp.metrics[k], rtol=1e-9), k
