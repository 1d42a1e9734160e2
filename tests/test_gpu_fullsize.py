"""Full-size (BASELINE.json configs[1]/[2]) checks on the GPU through properties that
do not need the CPU oracle at J = 65 536: agreement with an fp64 torch restatement of
the factored step evaluated on the device (checker only), shard additivity, noise
statistics, in-kernel vs injected noise.  p = n_obs = 256."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P, N, J = 256, 256, 65536


@pytest.fixture(scope="module")
def setup():
    import torch
    from ces_amd import build, engine
    build.build_lib()
    assert torch.cuda.is_available()
    rng = np.random.default_rng(20240)
    A = rng.standard_normal((N, P)) / np.sqrt(P)
    ustar = rng.standard_normal((P, 1))
    prob = dict(A=A, ustar=ustar, Gamma=0.01 * np.eye(N), y=(A @ ustar).ravel() + 0.1 * rng.standard_normal(N),
                mu=np.zeros((P, 1)), sigma=100.0 * np.eye(P))
    g = torch.Generator(device="cuda").manual_seed(7)
    U = torch.as_tensor(ustar, device="cuda") + torch.randn((P, J), generator=g, device="cuda", dtype=torch.float64)
    G = torch.as_tensor(A, device="cuda") @ U
    xi = torch.randn((P, J), generator=g, device="cuda", dtype=torch.float64)
    return engine, prob, U, G, xi


def torch_factored_aldi(prob, U, G, xi):
    """fp64 restatement of ces/calibrate.py:451-490 in J x J-free form, on device."""
    import torch
    dev = U.device
    T = lambda a: torch.as_tensor(a, device=dev, dtype=torch.float64)
    y, Gamma, mu, sigma, ustar = T(prob["y"]).reshape(-1, 1), T(prob["Gamma"]), T(prob["mu"]), T(prob["sigma"]), T(prob["ustar"])
    p, Jn = U.shape
    ubar, gbar = U.mean(1, keepdim=True), G.mean(1, keepdim=True)
    Au, E = U - ubar, G - gbar
    S_uu, S_ug, S_ee = Au @ Au.T, Au @ E.T, E @ E.T
    m = gbar - y
    S_rr = S_ee + Jn * (m @ m.T)
    Ginv = torch.linalg.inv(Gamma)
    frob = torch.sqrt(((Ginv @ S_rr @ Ginv.T) * S_ee).sum()) / Jn
    hk = 1.0 / (frob + 1e-8)
    C = S_uu / (Jn - 1) + 1e-8 * torch.eye(p, device=dev, dtype=torch.float64)
    L = torch.linalg.cholesky(C)
    K = (S_ug / Jn) @ Ginv
    M = C @ torch.linalg.inv(sigma)
    alpha = (p + 1.0) / Jn
    Uk = U - hk * (K @ (G - y)) - hk * (M @ (U - mu)) + hk * alpha * Au + torch.sqrt(2 * hk) * (L @ xi)
    R = G - y
    met = dict(self_bias=(torch.trace(S_uu) / Jn).item(),
               bias=(torch.trace(S_uu) / Jn + ((ubar - ustar) ** 2).sum()).item(),
               self_bias_data=((((Ginv @ E) * E).sum(0)) ** 2).mean().item(),
               bias_data=((((Ginv @ R) * R).sum(0)) ** 2).mean().item())
    return Uk, hk.item(), met


@pytest.mark.parametrize("dtype,tol", [("float32", 1e-3), ("float64", 1e-6)])
def test_c2_step_matches_fp64_restatement(setup, dtype, tol):
    import torch
    engine, prob, U, G, xi = setup
    eng = engine.Engine(P, N, J, dtype=dtype)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    Ud, Gd, xid = eng.to_device(U), eng.to_device(G), eng.to_device(xi)
    out = eng.step(engine.step_params(update="aldi"), Ud, Gd, xi=xid)
    res = eng.result()
    ref, hk, met = torch_factored_aldi(prob, Ud.double(), Gd.double(), xid.double())
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < tol, err
    mt = 1e-9 if dtype == "float64" else 1e-4
    assert res.hk == pytest.approx(hk, rel=mt)
    for k, v in met.items():
        assert getattr(res, k) == pytest.approx(v, rel=max(mt, 1e-8) * 10), k
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("update,kw,t_prev", [("aldi", {}, []), ("eks", {}, [0.3]), ("aldi_constant", {"switch": 0.5}, [0.3]),
                                              ("eks", {"time_step": "constant", "delta_t": 0.02}, [0.3]),
                                              ("aldi", {"time_step": "mix", "delta_t": 0.05, "spinup": 1.0}, [0.9, 1.5])])
def test_c2_fullsize_rules_match_pinned_oracle(setup, update, kw, t_prev):
    """J = 65 536 (config C2) against the PINNED CPU oracle (oracle.factored_step, fp64) for every update
    rule and the time-step rules that change the gain -- test_c2_step_matches_fp64_restatement and
    test_repeated_step_is_bit_identical are a third restatement and a self-comparison respectively;
    this is the independent check at full size.  fp32 engine, 1e-3 (north star)."""
    from oracle import ces_numpy as oc
    engine, prob, U, G, xi = setup
    Uh, Gh, xih = (t.to(dtype=__import__("torch").float32).cpu().numpy().astype(np.float64) for t in (U, G, xi))
    st = oc.OracleState(P, N, J, prob["mu"], prob["sigma"], prob["ustar"], T=30)
    st.metrics["t"] = list(t_prev)
    st.trace_len = 1 if not t_prev else 2
    ref = oc.factored_step(st, prob["y"], Uh, Gh, prob["Gamma"], xih, update=update, **kw)
    eng = engine.Engine(P, N, J, dtype="float32")
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    prm = engine.step_params(update=update, time_step=kw.get("time_step"), first_step=not t_prev, t_len=len(t_prev),
                             t_last=t_prev[-1] if t_prev else 0.0, delta_t=kw.get("delta_t"),
                             spinup=kw.get("spinup", 4.0), switch=kw.get("switch", 1.0), T=30)
    out = eng.step(prm, Uh, Gh, xi=xih)
    res = eng.result()
    got = out.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 1e-3
    assert res.t_new == pytest.approx(st.metrics["t"][-1], rel=1e-3)
    have = np.array([res.self_bias, res.self_bias_data, res.bias_data, res.bias])
    want = np.array([st.metrics[k][-1] for k in ("self-bias", "self-bias-data", "bias-data", "bias")])
    assert np.allclose(have, want, rtol=1e-3), (have, want)


def test_c3_eight_logical_shards_of_524288(setup):
    """Config C3's arithmetic at its real size on one device: J = 524 288 fp32 particles as 8 logical shards
    of 65 536 (what each of the 8 GPUs holds), moments summed on the device (stand-in for the RCCL
    all-reduce), apply per shard == the single-engine step over the whole ensemble."""
    import torch
    engine, prob, U, G, xi = setup
    Jg, nsh = 8 * J, 8
    g = torch.Generator(device="cuda").manual_seed(11)
    Ug = torch.as_tensor(prob["ustar"], device="cuda", dtype=torch.float32) + \
        torch.randn((P, Jg), generator=g, device="cuda", dtype=torch.float32)
    whole = engine.Engine(P, N, Jg, dtype="float32")
    whole.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    Gg = whole.forward_lineal(prob["A"], Ug)
    prm = engine.step_params(update="aldi", step_index=3)
    ref = whole.step(prm, Ug, Gg, xi=None)                  # on-device noise keyed by the global index
    rw = whole.result()
    shards = []
    for k in range(nsh):
        e = engine.Engine(P, N, J, dtype="float32", J_global=Jg, j_offset=k * J)
        e.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
        shards.append((e, Ug[:, k * J:(k + 1) * J].contiguous(), Gg[:, k * J:(k + 1) * J].contiguous()))
    sums = sum(e.colsum(Us, Gs) for e, Us, Gs in shards)
    for e, *_ in shards:
        e.set_shift(sums)
    mom = sum(e.moments(Us, Gs) for e, Us, Gs in shards)
    assert float(mom[0]) == Jg
    outs = [e.apply(prm, mom, Us, Gs, xi=None) for e, Us, Gs in shards]
    share = [e.result() for e, *_ in shards]
    got = torch.cat(outs, dim=1)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err
    assert all(s.hk == share[0].hk for s in share)          # every rank takes the same t_tol decision
    assert share[0].hk == pytest.approx(rw.hk, rel=1e-6)
    assert sum(s.bias_data for s in share) == pytest.approx(rw.bias_data, rel=1e-5)
    assert sum(s.self_bias_data for s in share) == pytest.approx(rw.self_bias_data, rel=1e-5)


def test_c5_per_gpu_shape_fp64():
    """Config C5 as one GPU holds it: fp64, p = n_obs = 512, J = 32 768, ALDI with the on-device blocked
    Cholesky, against the PINNED oracle (oracle.factored_step in fp64 on the host: ~10 s at this size; ensemble, hk and
    the data metric) and the fp64 torch restatement on the device (every metric; 1e-6, north star) -- plus the same
    problem at J = 4 096 against the pinned oracle."""
    import torch
    from ces_amd import engine
    from oracle import ces_numpy as oc
    p = n = 512
    rng = np.random.default_rng(1)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    prob = dict(A=A, ustar=ustar, Gamma=0.01 * np.eye(n), y=(A @ ustar).ravel() + 0.1 * rng.standard_normal(n),
                mu=np.zeros((p, 1)), sigma=100.0 * np.eye(p))
    g = torch.Generator(device="cuda").manual_seed(3)
    for Jn, check in ((32768, "torch+oracle"), (4096, "oracle")):
        U = torch.as_tensor(ustar, device="cuda") + torch.randn((p, Jn), generator=g, device="cuda", dtype=torch.float64)
        G = torch.as_tensor(A, device="cuda") @ U
        xi = torch.randn((p, Jn), generator=g, device="cuda", dtype=torch.float64)
        eng = engine.Engine(p, n, Jn, dtype="float64")
        eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
        out = eng.step(engine.step_params(update="aldi"), U, G, xi=xi)
        res = eng.result()
        if "torch" in check:
            ref, hk, met = torch_factored_aldi(prob, U, G, xi)
            assert ((out - ref).abs().max() / ref.abs().max()).item() < 1e-6
            assert res.hk == pytest.approx(hk, rel=1e-9)
            for k, v in met.items():
                assert getattr(res, k) == pytest.approx(v, rel=1e-7), k
            del ref
        if "oracle" in check:
            st = oc.OracleState(p, n, Jn, prob["mu"], prob["sigma"], prob["ustar"])
            ref = oc.factored_step(st, prob["y"], U.cpu().numpy(), G.cpu().numpy(), prob["Gamma"], xi.cpu().numpy())
            assert np.max(np.abs(out.cpu().numpy() - ref)) / np.max(np.abs(ref)) < 1e-6
            assert res.hk == pytest.approx(st.metrics["t"][-1], rel=1e-8)
            assert res.bias_data == pytest.approx(st.metrics["bias-data"][-1], rel=1e-8)
        del eng, U, G, xi, out
        torch.cuda.empty_cache()


def test_c5_eight_logical_shards_of_262144():
    """Config C5's arithmetic at its GLOBAL size on one device: J = 262 144 fp64 particles, p = n_obs = 512, as 8
    logical shards of 32 768 (what each of the 8 GPUs holds): moments summed on the device (stand-in for the RCCL
    all-reduce), the blocked on-device Cholesky of the 512 x 512 covariance run redundantly by every shard, apply
    per shard == the single-engine step over the whole ensemble (on-device noise keyed by the global index)."""
    import torch
    from ces_amd import engine
    p = n = 512
    Js, nsh = 32768, 8
    Jg = Js * nsh
    rng = np.random.default_rng(1)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    prob = dict(A=A, ustar=ustar, Gamma=0.01 * np.eye(n), y=(A @ ustar).ravel() + 0.1 * rng.standard_normal(n),
                mu=np.zeros((p, 1)), sigma=100.0 * np.eye(p))
    g = torch.Generator(device="cuda").manual_seed(17)
    Ug = torch.as_tensor(ustar, device="cuda") + torch.randn((p, Jg), generator=g, device="cuda", dtype=torch.float64)
    Gg = torch.as_tensor(A, device="cuda") @ Ug
    whole = engine.Engine(p, n, Jg, dtype="float64")
    whole.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    prm = engine.step_params(update="aldi", step_index=2)
    ref = whole.step(prm, Ug, Gg, xi=None)
    rw = whole.result()
    shards = []
    for k in range(nsh):
        e = engine.Engine(p, n, Js, dtype="float64", J_global=Jg, j_offset=k * Js)
        e.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
        shards.append((e, Ug[:, k * Js:(k + 1) * Js].contiguous(), Gg[:, k * Js:(k + 1) * Js].contiguous()))
    sums = sum(e.colsum(Us, Gs) for e, Us, Gs in shards)
    for e, *_ in shards:
        e.set_shift(sums)
    mom = sum(e.moments(Us, Gs) for e, Us, Gs in shards)
    assert float(mom[0]) == Jg
    outs = [e.apply(prm, mom, Us, Gs, xi=None) for e, Us, Gs in shards]
    share = [e.result() for e, *_ in shards]
    got = torch.cat(outs, dim=1)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-9, err                                  # (1e-6 is the north star's fp64 bar)
    assert all(s.hk == share[0].hk for s in share)          # every rank takes the same t_tol decision
    assert share[0].hk == pytest.approx(rw.hk, rel=1e-10)
    assert sum(s.bias_data for s in share) == pytest.approx(rw.bias_data, rel=1e-9)
    assert sum(s.self_bias_data for s in share) == pytest.approx(rw.self_bias_data, rel=1e-9)
    del whole, shards, outs, got, ref, Ug, Gg
    torch.cuda.empty_cache()


def test_c3_logical_shards_match_whole_ensemble(setup):
    """configs[2] arithmetic on one device: 4 column shards, moments summed on device
    (stand-in for the RCCL all-reduce), apply per shard == single-shard step."""
    import torch
    engine, prob, U, G, xi = setup
    whole = engine.Engine(P, N, J, dtype="float32")
    whole.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    prm = engine.step_params(update="aldi", step_index=3)
    ref = whole.step(prm, U, G, xi=None).clone()           # on-device noise keyed by the global index
    rw = whole.result()
    nsh = 4
    cuts = [J * k // nsh for k in range(nsh + 1)]
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        e = engine.Engine(P, N, b - a, dtype="float32", J_global=J, j_offset=a)
        e.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
        shards.append((e, e.to_device(U[:, a:b]), e.to_device(G[:, a:b])))
    sums = sum(e.colsum(Us, Gs) for e, Us, Gs in shards)
    for e, *_ in shards:
        e.set_shift(sums)
    mom = sum(e.moments(Us, Gs) for e, Us, Gs in shards)
    outs = [e.apply(prm, mom, Us, Gs, xi=None) for e, Us, Gs in shards]
    share = [e.result() for e, *_ in shards]
    got = torch.cat(outs, dim=1)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err
    assert share[0].hk == pytest.approx(rw.hk, rel=1e-6)
    # the shards' shares of the data metrics add up to the whole-ensemble values
    assert sum(s.bias_data for s in share) == pytest.approx(rw.bias_data, rel=1e-5)
    assert sum(s.self_bias_data for s in share) == pytest.approx(rw.self_bias_data, rel=1e-5)


def test_device_noise_statistics_full_size(setup):
    import torch
    engine, prob, U, G, xi = setup
    eng = engine.Engine(P, N, J, dtype="float32", seed=99)
    z = eng.draw_noise(11).double()
    assert abs(z.mean().item()) < 1e-3 and abs(z.var().item() - 1.0) < 2e-3
    assert abs((z ** 4).mean().item() - 3.0) < 0.02                      # kurtosis of N(0,1)
    c = (z[:, :8192] @ z[:, :8192].T) / 8192                              # rows are uncorrelated
    off = c - torch.diag(torch.diag(c))
    assert off.abs().max().item() < 0.08
    z2 = eng.draw_noise(12).double()
    assert abs((z * z2).mean().item()) < 1e-3                             # steps are independent


def test_in_kernel_noise_equals_injected_full_size(setup):
    engine, prob, U, G, xi = setup
    eng = engine.Engine(P, N, J, dtype="float32", seed=5)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    prm = engine.step_params(update="aldi", step_index=2)
    a = eng.step(prm, U, G, xi=None).clone()
    eng.result()
    b = eng.step(prm, U, G, xi=eng.draw_noise(2))
    eng.result()
    assert ((a - b).abs().max() / b.abs().max()).item() < 1e-5


def test_sharded_sampler_single_rank_chain(setup):
    """ShardedSampler (device-resident chain, device forward hook) == step-by-step drop-in class."""
    import torch
    from ces_amd.dist import ShardedSampler
    from ces_amd.utils import lineal
    engine, prob, U, G, xi = setup
    Js, T = 4096, 4
    eng = engine.Engine(P, N, Js, dtype="float64")
    smp = ShardedSampler(eng, P, N, Js)
    smp.T = T
    xis = xi[:, : Js * T].reshape(P, T, Js).permute(1, 0, 2).contiguous()
    Uf = smp.run(prob["y"], U[:, :Js], lineal(prob["A"]), prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"],
                 update="aldi", xis=xis, t_tol=1e9)
    Uc = U[:, :Js].double()
    A = torch.as_tensor(prob["A"], device="cuda")
    for i in range(T):
        Uc, hk, met = torch_factored_aldi(prob, Uc, A @ Uc, xis[i].double())
        assert smp.metrics["bias-data"][i] == pytest.approx(met["bias_data"], rel=1e-8)
    assert ((Uf - Uc).abs().max() / Uc.abs().max()).item() < 1e-8
    assert len(smp.metrics["t"]) == T


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-6), ("float32", 1e-3)])
def test_c5_dimensions_p512(dtype, tol):
    """BASELINE.json configs[4] dimensions (d = 512, n_obs = 512; ALDI, on-device Cholesky
    through the global-memory kernel, Gram through the rectangle partition) on one shard."""
    import torch
    from ces_amd import engine
    p = n = 512
    Jn = 4096
    rng = np.random.default_rng(1)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    prob = dict(A=A, ustar=ustar, Gamma=0.01 * np.eye(n), y=(A @ ustar).ravel() + 0.1 * rng.standard_normal(n),
                mu=np.zeros((p, 1)), sigma=100.0 * np.eye(p))
    g = torch.Generator(device="cuda").manual_seed(3)
    U = torch.as_tensor(ustar, device="cuda") + torch.randn((p, Jn), generator=g, device="cuda", dtype=torch.float64)
    G = torch.as_tensor(A, device="cuda") @ U
    xi = torch.randn((p, Jn), generator=g, device="cuda", dtype=torch.float64)
    eng = engine.Engine(p, n, Jn, dtype=dtype)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    Ud, Gd, xid = eng.to_device(U), eng.to_device(G), eng.to_device(xi)
    out = eng.step(engine.step_params(update="aldi"), Ud, Gd, xi=xid)
    res = eng.result()
    ref, hk, met = torch_factored_aldi(prob, Ud.double(), Gd.double(), xid.double())
    assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < tol
    assert res.hk == pytest.approx(hk, rel=1e-4 if dtype == "float32" else 1e-9)


@pytest.mark.parametrize("p", [260, 300, 384, 512, 700])
@pytest.mark.parametrize("update", ["aldi", "eks"])
def test_blocked_cholesky_large_p(p, update):
    """p > 256: chol(C) through the blocked factorisation (256-wide register Cholesky of the diagonal
    blocks, register TRSM of the rows below, fp64 GEMM trailing update) against numpy, for matrix
    sizes that are / are not multiples of the block and tile sizes, and the step that uses it
    (eks also factors Sigma + hk C through the same workspace)."""
    import torch
    from ces_amd import engine
    n, J = 40, 2048
    rng = np.random.default_rng(p)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    U = ustar + rng.standard_normal((p, J))
    G = A @ U
    eng = engine.Engine(p, n, J, dtype="float64")
    eng.set_problem((A @ ustar).ravel(), 0.01 * np.eye(n), np.zeros((p, 1)), 100.0 * np.eye(p), ustar)
    out = eng.step(engine.step_params(update=update, time_step="constant" if update == "eks" else None, delta_t=0.01),
                   eng.to_device(U), eng.to_device(G), xi=eng.to_device(rng.standard_normal((p, J))))
    res = eng.result()
    assert np.isfinite(res.hk) and bool(torch.isfinite(out).all())
    dd = eng.debug_dense()
    div = J if update == "eks" else J - 1
    Uc = U - U.mean(axis=1, keepdims=True)
    C = Uc @ Uc.T / div + 1e-8 * np.eye(p)
    assert np.abs(dd["C"] - C).max() / np.abs(C).max() < 1e-10
    L = np.linalg.cholesky(dd["C"])
    assert np.abs(np.tril(dd["L"]) - L).max() / np.abs(L).max() < 1e-9


@pytest.mark.parametrize("shape,update,reps", [((256, 256, 65536), "aldi", 40), ((300, 40, 50004), "aldi", 25),
                                               ((256, 256, 65536), "eks", 10), ((256, 256, 65536), "aldi_constant", 10)])
def test_repeated_step_is_bit_identical(shape, update, reps):
    """Race screen and reproducibility: the same step (same inputs, same Philox step index,
    recentred from the data every time) gives bit-identical ensembles and metrics on every
    repeat -- no float atomics anywhere, and the LDS-DMA ring of K3 / the side-stream Cholesky
    are correctly ordered (tools/soak_determinism.py runs the long version)."""
    import torch
    from ces_amd import engine
    p, n, J = shape
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    eng = engine.Engine(p, n, J, dtype="float32", seed=7)
    eng.set_problem((A @ ustar).ravel(), 0.01 * np.eye(n), np.zeros((p, 1)), 100.0 * np.eye(p), ustar)
    g = torch.Generator(device="cuda").manual_seed(1)
    U = torch.as_tensor(ustar, device="cuda", dtype=torch.float32) + torch.randn((p, J), generator=g, device="cuda")
    G = eng.forward_lineal(A, U)
    prm = engine.step_params(update=update, first_step=False, t_len=1, t_last=0.1, step_index=3)
    ref = None
    for _ in range(reps):
        out = eng.step(prm, U, G, xi=None, recenter=True)
        res = eng.result()
        key = (res.hk, res.bias_data, res.self_bias_data, res.self_bias)
        if ref is None:
            ref = (out.clone(), key)
        else:
            assert torch.equal(out, ref[0]) and key == ref[1]
