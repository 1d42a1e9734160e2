#!/usr/bin/env python3
"""Headline benchmark: EKS particle-updates/s (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- K1 moments, the all-reduce, K2 small
dense algebra, K3 fused update with on-device noise, and the host's read of
hk / metrics that the driver loop needs for its t_tol test
(ces/calibrate.py:387; the read of step i overlaps the Gram of step i+1, as in
ces_amd.dist.ShardedSampler.run) -- over one resident batch (U, G) of the synthetic
linear-Gaussian problem of SURVEY.md 8(d).  Workload at every N: config C2 per
GPU (J = 65 536 particles per GPU, p = n_obs = 256, fp32, ALDI, default
Frobenius time step), i.e. weak scaling; N = 8 is config C3.  Inputs are in HBM
before the timed region; a ring of 4 distinct batches (537 MB > the 256 MB
Infinity Cache) is cycled so that no step re-reads a cache-resident batch.
Prints ONE JSON line on rank 0.

``python bench.py --gpus N`` without a launcher starts the N ranks itself (a child
``python -m torch.distributed.run`` started BEFORE this process touches the GPU) and relays
rank 0's line.  ``--config C5`` runs BASELINE.json configs[4]'s per-GPU shape (fp64, p = n_obs =
512, J = 32 768) instead of the headline C2.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_PEAK_TF = {"float32": 157.3, "float64": 78.6}   # dense matrix peaks, f32-in / f64 MFMA


def cpu_share():
    """Host cores this process may actually use (ces_amd.engine.cpu_share: cgroup quota, else the affinity mask)."""
    from ces_amd.engine import cpu_share as _share
    return _share()


def limit_host_threads(reserve=0):
    """Size the BLAS / torch host thread pools to the CPU share minus `reserve` (returns the thread count).  The
    end-to-end host-array leg reserves the engine's copy threads: BLAS threads spin on after a GEMM, and BLAS + copy
    threads above the cgroup quota get the whole process throttled (ces_amd.engine.default_copy_threads)."""
    n = max(1, cpu_share() - reserve)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(n)
    except Exception:
        pass
    torch.set_num_threads(n)
    return n


def sysfs_sclk():
    """The shader-clock level sysfs marks as current (an instantaneous sample: between two kernels it may read a
    sleep state; the in-kernel clock is `clock.k3_ghz`)."""
    import glob
    for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            cur = [ln.strip() for ln in open(path) if "*" in ln]
            if cur:
                return cur[0]
        except OSError:
            pass
    return None


def synthetic_problem(p, n, seed=20240, dense_gamma=False, dense_sigma=False):
    """SURVEY.md 8(d) [decision]: A ~ N(0,1)/sqrt(p), Gamma = 0.01 I, mu = 0, Sigma = 100 I; the dense variants
    ("Gamma = L L^T + n I", scaled to the same size: 0.01 (B B^T / n + I), 100 (B B^T / p + I)) take the general path."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    Gamma = 0.01 * np.eye(n)
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    sigma = 100.0 * np.eye(p)
    rng2 = np.random.default_rng(seed + 1)          # (the diagonal problem's numbers do not move)
    if dense_gamma:
        B = rng2.standard_normal((n, n))
        Gamma = 0.01 * (B @ B.T / n + np.eye(n))
    if dense_sigma:
        B = rng2.standard_normal((p, p))
        sigma = 100.0 * (B @ B.T / p + np.eye(p))
    return dict(A=A, ustar=ustar, Gamma=Gamma, y=y, mu=np.zeros((p, 1)), sigma=sigma)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(prob, p, n, J, dtype, budget_s=10.0, max_steps=12):
    """The numpy CPU path on this box's host cores:
    * ``value``        oracle.factored_step in the engine's dtype at the full J (the J x J-free form is
                       the only one that fits in host memory at J = 65 536);
    * ``factored_f64`` the same in fp64 (the reference's own dtype);
    * ``literal``      oracle.literal_step = ces/calibrate.py:451-490 line by line (forms the J x J
                       matrix D), fp64, at J = 16 384 (2 x 8 J^2 = 4.3 GB); its cost grows as J^2, so the
                       rate at the benchmark's J is an EXTRAPOLATION (measured rate x 16384 / J), labelled."""
    from oracle import ces_numpy as oc
    limit_host_threads()
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    except Exception:
        cores = cpu_share()
    rng = np.random.default_rng(1)

    def leg(step, Jl, dt, budget, most):
        U = (prob["ustar"] + rng.standard_normal((p, Jl))).astype(dt)
        G = (prob["A"].astype(dt) @ U)
        st = oc.OracleState(p, n, Jl, prob["mu"], prob["sigma"], prob["ustar"])
        steps, t0 = 0, time.perf_counter()
        while steps < most and (steps == 0 or time.perf_counter() - t0 < budget):
            xi = np.random.normal(0, 1, [p, Jl])             # ces/calibrate.py:488 draws inside the update
            st.trace_len = 1 if steps == 0 else 2
            if step is oc.factored_step:
                step(st, prob["y"], U, G, prob["Gamma"], xi, update="aldi", dtype=dt)
            else:
                step(st, prob["y"], U, G, prob["Gamma"], xi, update="aldi")
            steps += 1
        el = time.perf_counter() - t0
        return Jl * steps / el, steps, el

    v, steps, el = leg(oc.factored_step, J, dtype, budget_s, max_steps)
    out = dict(value=v, unit="particle-updates/s", cores=int(cores), kind="port", cpu_model=_cpu_model(),
               sample="%d steps of oracle.factored_step (J x J-free numpy restatement of "
                      "ces/calibrate.py:451-490, %s, incl. np.random.normal) at J=%d, p=%d, n_obs=%d; "
                      "%.1f s" % (steps, np.dtype(dtype).name, J, p, n, el))
    if np.dtype(dtype) != np.dtype(np.float64):
        v64, s64, e64 = leg(oc.factored_step, J, np.float64, 0.6 * budget_s, max_steps)
        out["factored_f64"] = dict(value=v64, unit="particle-updates/s",
                                   sample="%d steps, float64, J=%d; %.1f s" % (s64, J, e64))
    Jl = min(J, 16384)
    vl, sl, elit = leg(oc.literal_step, Jl, np.float64, 0.5 * budget_s, 3)
    out["literal"] = dict(value=vl, unit="particle-updates/s", J=Jl,
                          sample="%d steps of oracle.literal_step (ces/calibrate.py:451-490 as written, forms "
                                 "the J x J matrix D), float64, J=%d; %.1f s" % (sl, Jl, elit),
                          extrapolated_to_J=J, extrapolated_value=vl * Jl / J,
                          extrapolation="O(J^2) per step: rate scaled by %d / %d; the literal form cannot run at "
                                        "J = %d (%.1f GB per J x J matrix)" % (Jl, J, J, 8.0 * J * J / 1e9))
    return out


def parity_check(engine, prob, p, n, dtype, update):
    """Max relative error of one step of the benchmarked configuration (same problem, same rule, injected
    xi) against the pinned CPU oracle in fp64, on a J = 4 096 ensemble (SURVEY.md 8d "also report")."""
    from oracle import ces_numpy as oc
    Jc = 4096
    rng = np.random.default_rng(7)
    U = prob["ustar"] + rng.standard_normal((p, Jc))
    G = prob["A"] @ U
    xi = rng.standard_normal((p, Jc))
    cast = (lambda a: a.astype(np.float32).astype(np.float64)) if np.dtype(dtype) == np.dtype(np.float32) else (lambda a: a)
    U, G, xi = cast(U), cast(G), cast(xi)
    st = oc.OracleState(p, n, Jc, prob["mu"], prob["sigma"], prob["ustar"])
    ref = oc.factored_step(st, prob["y"], U, G, prob["Gamma"], xi, update=update)
    eng = engine.Engine(p, n, Jc, dtype=dtype)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    out = eng.step(engine.step_params(update=update), U, G, xi=xi).cpu().numpy().astype(np.float64)
    res = eng.result()
    return dict(J=Jc, U_next_max_rel=float(np.max(np.abs(out - ref)) / np.max(np.abs(ref))),
                hk_rel=float(abs(res.hk - st.metrics["t"][-1]) / st.metrics["t"][-1]),
                bias_data_rel=float(abs(res.bias_data - st.metrics["bias-data"][-1]) / st.metrics["bias-data"][-1]),
                against="oracle.factored_step float64 (pinned to the reference through tests/golden)",
                bar=1e-3 if np.dtype(dtype) == np.dtype(np.float32) else 1e-6)


def e2e_block(engine, prob, p, n, J, dtype, update, dev_index):
    """End-to-end rates of the two ways a user drives the engine (SURVEY.md 8d "also report"; never `value`):
    ``device_chain``  the ensemble stays in HBM for the whole run: forward map G = A U on the device every
                      step (lineal.forward_device), update, U_next fed back (ShardedSampler.run);
    ``host_arrays``   the reference's calling convention: float64 numpy U and G in, float64 numpy U_next out
                      (sampling.eks_update_aldi), G = A U by host BLAS -- PCIe both ways every step."""
    from ces_amd.calibrate import sampling
    from ces_amd.dist import ShardedSampler
    from ces_amd.utils import lineal
    out = {}
    nthreads = limit_host_threads(reserve=engine.Engine.copy_threads)
    rng = np.random.default_rng(3)
    U0 = prob["ustar"] + rng.standard_normal((p, J))
    model = lineal(prob["A"])
    eng = engine.Engine(p, n, J, dtype=dtype, device=dev_index, seed=77)
    for T, timed, fast in ((8, False, True), (200, True, True), (8, False, False), (200, True, False)):
        # fast: the linear map lives in the engine, so the G-dependent moments follow from the U-only head
        # (cesx_moments_rest_lineal) and the forward GEMM runs beside chol(C); not fast: forward map, then the full Gram
        os.environ["CESX_LINEAL_FAST"] = "1" if fast else "0"
        smp = ShardedSampler(eng, p, n, J)
        smp.T = T
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smp.run(prob["y"], U0, model, prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"], update=update,
                t_tol=1e30)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if timed:
            key = "device_chain" if fast else "device_chain_full_gram"
            out[key] = dict(value=J * T / el, unit="particle-updates/s", ms_per_step=1e3 * el / T, steps=T,
                            lineal_fast_path=bool(fast and smp.sh.lineal_fast_ok(model)),
                            includes="upload of U0 once (67 MB at C2: ~3 ms, spread over the steps), then per step: G = A U on device, update with "
                                     "on-device noise, host read of t; U_next fed back" +
                                     ("; the G-dependent moments from the U-only head and the installed linear map "
                                      "(no second Gram launch)" if fast else "; full Gram over [U; G]"))
    os.environ.pop("CESX_LINEAL_FAST", None)
    del eng
    # the reference's own calling convention: sampling.run with a host forward map and float64 numpy arrays
    # across PCIe into and out of every update (device_loop off).  G_ens evaluates the linear map for the whole
    # ensemble with one host GEMM (the reference's per-particle Python loop would take ~0.1 s per step on its own)
    class host_lineal:
        type, model_name, n_obs = "map", "lineal", n

        def __call__(self, theta):
            return prob["A"] @ theta
    fwd_ms, stamps = [], []

    def g_ens(theta, m):
        t0 = time.perf_counter()
        g = prob["A"] @ theta
        t1 = time.perf_counter()
        fwd_ms.append(1e3 * (t1 - t0))
        stamps.append((t0, t1))
        return g
    nst = 24

    def cpu_stat():
        try:
            return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
        except Exception:
            return {}
    eks = sampling(p=p, n_obs=n, J=J)
    eks.mu, eks.sigma, eks.ustar = prob["mu"], prob["sigma"], prob["ustar"]
    eks.engine_dtype, eks.noise, eks.device, eks.device_loop = np.dtype(dtype).name, "device", dev_index, False
    eks.G_ens = g_ens
    eks.T = 3
    eks.run(prob["y"], U0, host_lineal(), prob["Gamma"], None, trace=False, t_tol=1e30)      # builds engine + pinned buffers
    eks.T = nst
    del fwd_ms[:], stamps[:]
    st0 = cpu_stat()
    t0 = time.perf_counter()
    eks.run(prob["y"], eks.Ustar, host_lineal(), prob["Gamma"], None, trace=False, t_tol=1e30)
    el = time.perf_counter() - t0
    st1 = cpu_stat()
    # The pipelined host loop (sampling._run_host_pipelined) evaluates the forward map block by block -- several
    # G_ens calls per iteration plus ONE call on the final ensemble.  An iteration = from the start of its first
    # block's forward evaluation to the start of the next iteration's; the final whole-ensemble call is left out.
    per_it = (len(stamps) - 1) // nst if len(stamps) > nst else 1          # forward calls per iteration
    starts = [stamps[k * per_it][0] for k in range(nst)] + [stamps[nst * per_it][0]] if len(stamps) > nst * per_it else []
    iters = [1e3 * (starts[k + 1] - starts[k]) for k in range(len(starts) - 1)]
    fwd_it = [sum(fwd_ms[k * per_it:(k + 1) * per_it]) for k in range(nst)]
    fwd = float(np.mean(fwd_it))
    calls = [iters[k] - fwd_it[k] for k in range(len(iters))]              # what an iteration spends outside the forward map
    el = (starts[-1] - starts[0]) if starts else el
    pct = lambda v: dict(p50=round(float(np.percentile(v, 50)), 3), p95=round(float(np.percentile(v, 95)), 3),
                         max=round(float(np.max(v)), 3)) if len(v) else None
    out["host_arrays"] = dict(value=J * nst / el, unit="particle-updates/s", steps=nst, ms_per_step=1e3 * el / nst,
                              host_forward_ms=fwd, update_call_ms=1e3 * el / nst - fwd,
                              host_forward_ms_median=float(np.median(fwd_it)),
                              update_call_ms_median=float(np.median(calls)) if calls else None,
                              iteration_ms_pct=pct(iters), update_call_ms_pct=pct(calls), host_forward_ms_pct=pct(fwd_it),
                              forward_calls_per_iteration=per_it,
                              # where an iteration's time outside the forward map goes (ms per step): the driving thread's
                              # waits (down0 / down1: first / later blocks of the new ensemble not yet widened; up: G blocks
                              # not yet cast and enqueued; result: the step's scalars) and the staging thread's own work
                              driver_waits_ms={k: round(1e3 * v / nst, 3) for k, v in sorted(getattr(eks, "_pipe_times", {}).items())},
                              stager_work_ms={k: round(1e3 * v / nst, 3) for k, v in sorted(getattr(eks, "_pipe_stage_times", {}).items())},
                              host_threads=nthreads, copy_threads=int(engine.Engine.copy_threads),
                              cgroup_periods_throttled="%d of %d" % (st1.get("nr_throttled", 0) - st0.get("nr_throttled", 0),
                                                                     st1.get("nr_periods", 0) - st0.get("nr_periods", 0)),
                              includes="sampling.run(trace=False) with a host forward map (numpy A @ U) and float64 numpy "
                                       "arrays, pipelined over 4 column blocks of the ensemble: block c of the new ensemble comes "
                                       "down, G_ens evaluates it, its G goes up while the next block is evaluated; update_call = "
                                       "what an iteration spends outside the forward map; PCIe Gen5 floor for the "
                                       "%.0f MB of engine-dtype traffic per step ~%.1f ms; BLAS threads + the engine's "
                                       "copy threads = the cgroup CPU share (more gets the process throttled)"
                                       % ((p + n) * J * np.dtype(dtype).itemsize / 1e6,
                                          (p + n) * J * np.dtype(dtype).itemsize / 56e9 * 1e3))
    return out


def flops_executed(p, n, J, dtype, form=1):
    """MFMA flops a step really issues (what SQ_VALU_MFMA_BUSY_CYCLES counts), next to the algorithmic ones of SURVEY.md
    8(d): K1 computes whole 32 x 32 (fp32) / 16 x 16 (fp64) blocks of the lower triangle of Z Z^T -- the diagonal blocks'
    upper halves are computed and thrown away --, K3 skips the zero blocks of its lower-triangular sqrt(2hk) L segment at
    the granularity of a row block (32 / 16 rows): 2 p J (p + n) + p J (p + row block).  form 2 (cesx_debug_update_form:
    K3 through the Cholesky factor, round 6): TWO block-triangular products + the dense K G: (nb (nb + 1) + nb ng / 2)
    products of 32 x 32 x 32 per 32 particles, nb = ceil(p / 32), ng = ceil(n / 16)."""
    tile = 32 if np.dtype(dtype) == np.dtype(np.float32) else 16
    nbr = -(-(p + n) // tile)
    k1 = 2.0 * (nbr * (nbr + 1) // 2) * tile * tile * J
    k3 = 2.0 * p * J * (p + n) + 1.0 * p * J * (p + tile)
    if form == 2:
        nb, ng = -(-p // 32), -(-n // 16)
        k3 = 2.0 * 32 ** 3 * (nb * (nb + 1) + nb * ng / 2.0) * (J / 32.0)
    return k1, k3


def k3_flops_algorithmic(p, n, J, form=1):
    """K3's algorithmic flops per launch.  SURVEY.md 8(d) prices the fused update GEMM at 2 p (2p + n) per particle -- the dense
    C Sigma^{-1} U product (2 p^2), the gain term (2 p n) and the dense-counted L xi (2 p^2).  Through the Cholesky factor (form 2)
    the same update is L (sqrt(2/hk) xi - L^T Sigma^{-1} U): two TRIANGULAR products, p (p + 1) flops each, and the gain term --
    2 p (p + 1) + 2 p n per particle.  Pricing the faster kernel with the survey's larger count would put it above the peak
    (25.8 GFLOP in 0.155 ms = 166 TF of a 157-TF pipe); the line therefore prices what the algorithm now needs and carries
    the survey's figure beside it (roofline.kernels[K3].flops_survey_8d)."""
    if form == 2:
        return (2.0 * p * (p + 1) + 2.0 * p * n) * J
    return 2.0 * p * (2 * p + n) * J


def engine_leg(engine, name, p, n, J, dtype, steps, dev_index, prewarm_s=0.6, update="aldi", time_step=None,
               dense_gamma=False, dense_sigma=False, t_start=0.0, sample=True):
    """A short engine-only leg of another BASELINE.json configuration on one GPU, measured like the headline (ring of 4
    resident batches, pipelined begin / finish / result, continuous stepping up to the timed region, one
    HIP-event-sampled step in its middle): ms/step, particle-updates/s and the K1 / K3 roofline fractions."""
    from ces_amd.dist import ShardedUpdate
    dev = torch.device("cuda", dev_index)
    prob = synthetic_problem(p, n, dense_gamma=dense_gamma, dense_sigma=dense_sigma)
    eng = engine.Engine(p, n, J, dtype=dtype, device=dev_index, seed=1234)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    sh = ShardedUpdate(eng)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4242)
    ustar_d = torch.as_tensor(prob["ustar"], device=dev, dtype=eng.torch_dtype)
    batches = []
    for b in range(4):
        U = ustar_d + (1.0 + 0.05 * b) * torch.randn((p, J), generator=gen, device=dev, dtype=eng.torch_dtype)
        batches.append((U, eng.forward_lineal(prob["A"], U)))
    out = eng.empty(p)
    prm0 = engine.step_params(update=update)
    t_hist, at = [t_start], [-1]
    eng.profile_enable(True)
    eng.profile_enable(False)

    def begin(i):
        U, G = batches[i % 4]
        eng.profile_enable(i == at[0])
        sh.begin(prm0, U, G, recenter=(i == 0), noise_step=i)

    def finish(i):
        U, G = batches[i % 4]
        eng.profile_enable(i == at[0])
        # (t_start > 0: a run that is already past its spin-up -- time_step='mix' then takes delta_t and recomputes the gain)
        first = i == 0 and t_start == 0.0
        prm = engine.step_params(update=update, time_step=time_step, first_step=first, t_len=0 if first else 1,
                                 t_last=t_hist[0], step_index=i)
        sh.finish(prm, U, G, xi=None, out=out)

    def run_steps(first, count):
        begin(first)
        for i in range(first, first + count):
            finish(i)
            if i + 1 < first + count:
                begin(i + 1)
            # (the pseudo-time restarts where it began every 64 steps: a leg is thousands of steps of ONE ring of batches)
            t_hist[0] = eng.result().t_new if (i + 1) % 64 else t_start
    at[0] = 4 if sample else -1                  # first time-stamped launches: outside the timed region
    run_steps(0, 16)
    eng.profile_read(0), eng.profile_read(1)
    at[0] = -1
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < prewarm_s:
        run_steps(0, 64)
    torch.cuda.synchronize()
    at[0] = 3 + steps // 2 if sample else -1
    t0 = time.perf_counter()
    run_steps(3, steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eng.profile_enable(False)
    ghz = eng.profile_clock()
    g_ms, g_cnt = eng.profile_read(0)
    u_ms, u_cnt = eng.profile_read(1)
    peak = MFMA_PEAK_TF[np.dtype(dtype).name]
    form = eng.update_form()
    k1f, k3f = float(p + n) ** 2 * J, k3_flops_algorithmic(p, n, J, form)
    k1x, k3x = flops_executed(p, n, J, dtype, form)
    kern = {"gram_kernel(K1)": dict(avg_launch_ms=round(g_ms, 4), launches_per_step=g_cnt,
                                    tflops=round(k1f / (g_ms * 1e-3) / 1e12, 2) if g_ms > 0 else None,
                                    frac=round(k1f / (g_ms * 1e-3) / 1e12 / peak, 4) if g_ms > 0 else None,
                                    frac_executed=round(k1x / (g_ms * 1e-3) / 1e12 / peak, 4) if g_ms > 0 else None),
            "update_kernel(K3)": dict(avg_launch_ms=round(u_ms, 4), launches_per_step=u_cnt,
                                      tflops=round(k3f / (u_ms * 1e-3) / 1e12, 2) if u_ms > 0 else None,
                                      frac=round(k3f / (u_ms * 1e-3) / 1e12 / peak, 4) if u_ms > 0 else None,
                                      frac_executed=round(k3x / (u_ms * 1e-3) / 1e12 / peak, 4) if u_ms > 0 else None,
                                      update_form=form)}
    esz = np.dtype(dtype).itemsize
    rec = dict(workload="%s: J=%d, d=p=%d, n_obs=%d, %s, update=%s, time_step=%s, %s Gamma, %s Sigma, synthetic linear-Gaussian "
                        "inputs resident in HBM, on-device noise, engine only"
                        % (name, J, p, n, np.dtype(dtype).name, update, time_step, "dense" if dense_gamma else "diagonal",
                           "dense" if dense_sigma else "diagonal"),
               value=J * steps / el, unit="particle-updates/s", steps=steps, ms_per_step=1e3 * el / steps,
               dtype={"float32": "f32", "float64": "f64"}[np.dtype(dtype).name],
               roofline=dict(bound="mfma", peak=peak, unit="TFLOP/s", kernels=kern,
                             step_flops_frac=round((k1f + k3f) / (el / steps) / 1e12 / peak, 4),
                             step_algorithmic_bytes=float(esz * (3 * p + 2 * n)) * J,
                             step_algorithmic_hbm_frac=round(esz * (3 * p + 2 * n) * J / (el / steps) / 1e9 / HBM_PEAK_GBS, 4)),
               k3_clock_ghz=round(ghz, 4) if ghz else None)
    del batches, out, sh, eng
    torch.cuda.empty_cache()
    return rec


def darcy_leg(engine, dev_index, J=512, T=3):
    """BASELINE.json configs[3] end to end at a J the host finishes in seconds: the host Darcy forward map
    (ces_amd/darcy.py, model_trunc(p=64), 50 observations; the reference's MATLAB map restated, parity unpinned)
    + the GPU update, driven like examples/scripts/darcy-flow.py:43-93 through sampling.run."""
    from ces_amd import darcy
    from ces_amd.calibrate import sampling
    full = darcy.model(); full.set_initial(); full.n_obs = 50
    Ufull = full(full.ustar, full_solution=True)
    rng = np.random.RandomState(1)
    obs_index = rng.choice(int(full.p), 50, replace=False, p=Ufull / Ufull.sum())
    model = darcy.model_trunc(p=64); model.set_initial(); model.n_obs = 50; model.obs_index = obs_index
    gamma = 0.005
    Gamma = gamma ** 2 * np.identity(50)
    y_obs = model(model.ustar) + gamma * rng.normal(0, 1, 50)
    eks = sampling(p=model.p, n_obs=model.n_obs, J=J)
    eks.ustar = model.ustar.reshape(model.p, -1)
    eks.mu, eks.sigma = np.zeros((model.p, 1)), 100.0 * np.identity(model.p)
    eks.engine_dtype, eks.noise, eks.device, eks.T = "float32", "device", dev_index, T
    fwd, upd = [], []
    g_ens = eks.G_ens

    def timed_g_ens(theta, m):
        t0 = time.perf_counter()
        g = g_ens(theta, m)
        fwd.append(time.perf_counter() - t0)
        return g
    eks.G_ens = timed_g_ens
    upd_fn = eks.eks_update_aldi

    def timed_update(*a, **k):
        t0 = time.perf_counter()
        r = upd_fn(*a, **k)
        upd.append(time.perf_counter() - t0)
        return r
    eks.eks_update_aldi = timed_update
    U0 = 10 * rng.normal(0, 1, [eks.p, J])
    t0 = time.perf_counter()
    eks.run(y_obs, U0, model, Gamma, np.linalg.cholesky(Gamma), t_tol=1e30, trace=False)
    el = time.perf_counter() - t0
    # (the first call creates the engine and factors Gamma / Sigma: the steady-state call is the last one)
    fwd_s, upd_ms, upd_first_ms = float(np.mean(fwd)), float(upd[-1]) * 1e3, float(upd[0]) * 1e3
    return dict(J=J, iterations=len(upd), wall_s=round(el, 3), host_forward_s_per_iteration=round(fwd_s, 4),
                host_forward_ms_per_particle=round(1e3 * fwd_s / J, 4), update_call_ms=round(upd_ms, 4),
                first_update_call_ms=round(upd_first_ms, 4),
                forward_share=round(sum(fwd) / el, 4),
                extrapolated_J8192=dict(host_forward_s_per_iteration=round(fwd_s / J * 8192, 2),
                                        note="one host core, serial G_ens (ces/calibrate.py:123-130); the update of the "
                                             "J=8192 ensemble is the engine-only ms_per_step above"),
                includes="sampling.run(trace=False): per iteration one host forward evaluation of every particle "
                         "(sparse solve of the 5-point Darcy operator), float64 arrays across PCIe into and out of the "
                         "update (host-array calling convention), the final forward evaluation")


VARIANTS = (   # (key, kwargs of engine_leg) -- every rule / problem class of the same reference function (ces/calibrate.py:418-529)
    ("aldi_default", dict()),
    ("eks", dict(update="eks")),
    ("aldi_constant", dict(update="aldi_constant")),
    ("time_step_constant", dict(time_step="constant")),
    ("time_step_mix_late", dict(time_step="mix", t_start=5.0)),
    ("time_step_spectral", dict(time_step="spectral")),
    ("dense_gamma", dict(dense_gamma=True)),
    ("dense_sigma", dict(dense_sigma=True)),
    ("dense_gamma_sigma", dict(dense_gamma=True, dense_sigma=True)),
    ("eks_dense_gamma_sigma", dict(update="eks", dense_gamma=True, dense_sigma=True)),
)


def variants_leg(engine, p, n, J, dtype, dev_index, steps=10):
    """"a dense-Gamma / dense-Sigma variant is run once to show the general path costs the same" -- and the
    other update rules and time-step rules of the same reference function: engine-only legs of `steps` steps each at the
    headline shape, measured like the headline (pipelined begin / finish / result over 4 resident batches, continuous
    stepping up to the timed steps), with their ratio to the default step measured the same way."""
    out = {}
    for key, kw in VARIANTS:
        try:
            r = engine_leg(engine, "variant " + key, p, n, J, dtype, steps, dev_index, prewarm_s=0.25, sample=False, **kw)
            out[key] = dict(ms_per_step=round(r["ms_per_step"], 4), value=r["value"], **{k: v for k, v in kw.items()})
        except Exception as ex:                     # one leg must not cost the line the others
            out[key] = dict(error=repr(ex))
    base = out.get("aldi_default", {}).get("ms_per_step")
    for key, r in out.items():
        if base and "ms_per_step" in r:
            r["ratio_to_default"] = round(r["ms_per_step"] / base, 3)
    try:
        out["in_a_run"] = variants_in_a_run(engine, p, n, J, dtype, dev_index)
    except Exception as ex:
        out["in_a_run"] = dict(error=repr(ex))
    out["how"] = ("engine-only, %d timed steps each behind 0.25 s of continuous stepping, same shape and dtype as the headline; "
                  "time_step_mix_late: pseudo-time past the spin-up (delta_t and the recomputed gain of ces/calibrate.py:470-473)"
                  % steps)
    return out


def variants_in_a_run(engine, p, n, J, dtype, dev_index, T=120):
    """The rules that carry an hk-dependent SPD inverse (`eks`, the gain-recomputing time steps), measured INSIDE A RUN: the
    ensemble evolves (G = A U on the device, update, U_next fed back -- ShardedSampler.run, full Gram over [U; G]), so the
    previous step's inverse is the close start it is in use; the ring of four unrelated ensembles of the legs above is the
    warm start's worst case.  Ratio to the default ALDI step driven the same way.  (`mix`: the adaptive rule up to pseudo-time 1,
    then delta_t and the recomputed gain, ces/calibrate.py:256-260 / :470-473 -- about a third of the steps are default ones.)"""
    from ces_amd.dist import ShardedSampler
    from ces_amd.utils import lineal
    prob = synthetic_problem(p, n)
    rng = np.random.default_rng(3)
    U0 = prob["ustar"] + rng.standard_normal((p, J))
    model = lineal(prob["A"])
    out, dt = {}, None
    os.environ["CESX_LINEAL_FAST"] = "0"
    # host thread pools inside the cgroup's CPU share (a process above it is throttled, and a throttled driving thread shows as
    # a leg at twice the time -- tools/inrun_probe.py caught exactly that), the start ensemble uploaded once, outside the timing
    limit_host_threads(reserve=engine.Engine.copy_threads)
    U0 = torch.as_tensor(U0, dtype={"float32": torch.float32, "float64": torch.float64}[np.dtype(dtype).name]).to(torch.device("cuda", dev_index))
    try:
        for key, kw in (("aldi_default", dict(update="aldi")), ("eks", dict(update="eks")),
                        ("time_step_constant", dict(update="aldi", time_step="constant")),
                        ("time_step_mix", dict(update="aldi", time_step="mix", spinup=1.0))):
            eng = engine.Engine(p, n, J, dtype=dtype, device=dev_index, seed=77)
            try:
              el = None
              for steps, timed in ((8, False), (T, True), (T, True)):          # (the better of two timed runs)
                smp = ShardedSampler(eng, p, n, J)
                smp.T = steps
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                smp.run(prob["y"], U0, model, prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"], t_tol=1e30,
                        **(dict(kw, delta_t=dt) if "time_step" in kw else kw))
                torch.cuda.synchronize()
                if timed:
                    el = min(el, time.perf_counter() - t0) if el is not None else time.perf_counter() - t0
            except Exception as ex:
                out[key] = dict(error=repr(ex), steps_done=len(smp.metrics["t"]))
                continue
            out[key] = dict(ms_per_step=round(1e3 * el / T, 4), steps=T, last_inverse_warm=bool(eng.warm_inverse()))
            if key == "aldi_default":          # the fixed step of the constant / mix rules: half the adaptive rule's last one
                t = smp.metrics["t"]
                dt = 0.5 * float(t[-1] - t[-2])
            elif "time_step" in kw:
                out[key]["delta_t"] = dt
            del eng, smp
            torch.cuda.empty_cache()
    finally:
        os.environ.pop("CESX_LINEAL_FAST", None)
    base = out["aldi_default"].get("ms_per_step")
    for r in out.values():
        if base and "ms_per_step" in r:
            r["ratio_to_default"] = round(r["ms_per_step"] / base, 3)
    out["how"] = ("ShardedSampler.run, %d steps behind 8 untimed ones (the better of two runs): forward map on the device, full Gram, "
                  "update, feedback; the start ensemble is resident" % T)
    return out


def sharded_helper_main():
    """``bench.py --sharded-helper``: started by the one-GPU benchmark BEFORE it touches the GPU (a process that has
    initialised the GPU must not start another program); waits for a line on stdin, then runs the one-rank rehearsal of
    the SHARDED path -- RCCL really initialised, the step's all-reduces really issued (CESX_FORCE_COLLECTIVES=1) -- once
    per collective mode, each in a fresh child, and prints one JSON object."""
    if not sys.stdin.readline().strip():
        return                                       # parent went away / did not ask
    out = {}
    # (the third leg leaves the mode to the benchmark: 64 untimed steps in each behind the pre-warm, the faster one kept --
    #  what the driver's N > 1 runs do; config.parallelism and sampled_step.mode_choice say which and why)
    for key, extra_env in (("head_tail", {"CESX_SINGLE_ALLREDUCE": "0"}), ("single", {"CESX_SINGLE_ALLREDUCE": "1"}), ("auto", {})):
        env = dict(os.environ)
        env.update(CESX_FORCE_COLLECTIVES="1", CESX_BENCH_PREWARM_S="1.0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(sk.getsockname()[1])
        env["MASTER_ADDR"] = "127.0.0.1"
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--steps", "20",
                                  "--warmup", "3"], env=env, capture_output=True, text=True, timeout=240)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(line[-1])
            out[key] = dict(value=d["value"], ms_per_step=d["ms_per_step"], ms_per_step_median=d["ms_per_step_median"],
                            rccl_nranks=d["rccl_nranks"], rccl_comms=d.get("rccl_comms"), collective_mode=d["sampled_step"]["collective_mode"],
                            collectives_per_step=d["sampled_step"]["collectives_per_step"],
                            collective_ms=d["sampled_step"]["collective_ms"],
                            gram_end_to_k3_start_ms=d["sampled_step"]["gram_end_to_k3_start_ms"],
                            mode_choice=d["sampled_step"].get("mode_choice"),
                            parallelism=d["config"]["parallelism"])
        except Exception as ex:
            out[key] = dict(error=repr(ex))
    print(json.dumps(out), flush=True)


class Platform:
    """What main() needs of the machine.  The default is the GPU box (one rank per GPU over RCCL).  tests/bench_rehearsal.py
    substitutes a CPU stand-in -- gloo, the numpy engine of the CPU tests -- to drive the RANK CONTROL FLOW of this file
    (the pre-warm decision every rank must take alike, the barriers, the max-reduce of the elapsed time, the tear-down of
    the ranks that do not print) on two processes here, so that the driver's 8-GPU run does not meet it for the first time."""
    backend = "nccl"

    def device(self, local):
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def sync(self):
        torch.cuda.synchronize()

    def modules(self):
        from ces_amd import build, engine
        from ces_amd.dist import ShardedUpdate
        return build.build_lib, engine, ShardedUpdate

    def generator(self, dev):
        return torch.Generator(device=dev)


PLATFORM = Platform()


def small_j_leg(engine, dev_index, shapes=((64, 50, 512), (256, 50, 768)), calls=40):
    """The ensemble sizes the reference's own scripts run (examples/scripts/darcy-flow.py:97-105, darcy-flow.ipynb: J in
    17 .. 768 at p = 64 / 256): one ``eks_update_aldi`` call through the reference's calling convention -- float64 host
    arrays in, a fresh float64 host array out -- timed in steady state (median of `calls` after a few untimed ones),
    with the noise drawn on the device and from numpy's global stream (what the reference's CPU path pays too), next
    to the CPU restatement of ces/calibrate.py:451-490 as written (the J x J form) timed on this box (cpu_baseline leg)."""
    from ces_amd.calibrate import sampling
    out = []
    for (p, n, J) in shapes:
        rng = np.random.default_rng(5)
        A = rng.standard_normal((n, p)) / np.sqrt(p)
        ustar = rng.standard_normal((p, 1))
        Gamma, sigma, mu = 0.01 * np.eye(n), 100.0 * np.eye(p), np.zeros((p, 1))
        y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
        U0 = ustar + rng.standard_normal((p, J))
        G0 = A @ U0
        rec = dict(p=p, n_obs=n, J=J)
        for noise in ("device", "numpy"):
            eks = sampling(p=p, n_obs=n, J=J)
            eks.ustar, eks.mu, eks.sigma = ustar, mu, sigma
            eks.engine_dtype, eks.noise, eks.device, eks.T = "float32", noise, dev_index, 30
            eks.Uall = [U0, U0]                  # (not the first step: ces/calibrate.py:262)
            eks._ensure_metrics()
            eks.metrics["t"].append(0.1)
            tt = []
            for it in range(calls + 5):
                t0 = time.perf_counter()
                eks.eks_update_aldi(y, U0, G0, Gamma, it)
                if it >= 5:
                    tt.append(time.perf_counter() - t0)
            rec["update_call_ms_%s_noise" % noise] = round(float(np.median(tt)) * 1e3, 4)
        rec["cpu_literal_step_ms"] = cpu_baseline_small(p, n, J, mu, sigma, ustar, y, U0, G0, Gamma)
        rec["speedup_vs_cpu_literal"] = round(rec["cpu_literal_step_ms"] / rec["update_call_ms_numpy_noise"], 1)
        out.append(rec)
    return dict(cases=out, cores=cpu_share(),
                how="update_call_ms: median wall time of sampling.eks_update_aldi(y, U0, Geval, Gamma, i) with float64 numpy "
                    "arrays in and out (set-up unchanged between calls, the engine's problem fingerprint skips it); "
                    "cpu_literal_step_ms: oracle.literal_step -- ces/calibrate.py:451-490 as written, J x J matrices, numpy "
                    "on this box's host cores, its noise block drawn outside the timed call; speedup: against the numpy-noise "
                    "call, the like-for-like one (both sides draw p x J normals from numpy's global stream)")


def cpu_baseline_small(p, n, J, mu, sigma, ustar, y, U0, G0, Gamma, reps=5):
    """CPU baseline leg of small_j_leg: the oracle's literal step (test infrastructure, the CHECKER timed as the CPU
    reference -- never part of the measured path)."""
    from oracle import ces_numpy as oc
    st = oc.OracleState(p, n, J, mu, sigma, ustar)
    st.metrics["t"].append(0.1)
    rng = np.random.default_rng(6)
    tl = []
    for _ in range(reps):
        xi = rng.standard_normal((p, J))
        t0 = time.perf_counter()
        oc.literal_step(st, y, U0, G0, Gamma, xi, update="aldi")
        tl.append(time.perf_counter() - t0)
    return round(float(np.median(tl)) * 1e3, 3)


def self_launch(args):
    """No launcher (WORLD_SIZE unset) and --gpus N > 1: start the N ranks as children of this process, which
    has not touched the GPU, relay their output, exit with their status."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--J", type=int, default=65536, help="particles per GPU")
    ap.add_argument("--p", type=int, default=256)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--update", default="aldi")
    ap.add_argument("--config", default="C2", choices=["C2", "C5"],
                    help="C2: fp32, p=n=256, J=65536/GPU (headline); C5: fp64, p=n=512, J=32768/GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip parity_err and the e2e block")
    args = ap.parse_args()
    if args.config == "C5":
        args.J, args.p, args.n, args.dtype = 32768, 512, 512, "float64"

    # the HSA / RCCL environment is fixed before anything can initialise the runtime
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    plat = PLATFORM
    rehearse = world == 1 and os.environ.get("CESX_FORCE_COLLECTIVES") == "1"   # one-rank run of the N > 1 code path
    # The one-rank rehearsal of the sharded path (`extra.sharded_one_rank`) runs in fresh processes; their launcher is
    # started HERE, before this process touches the GPU, and sleeps on a pipe until the headline is measured.
    helper = None
    if (world == 1 and not rehearse and not args.no_extras and args.config == "C2" and plat.backend == "nccl"
            and os.environ.get("CESX_BENCH_NO_SHARDED_LEG") != "1"):
        try:
            helper = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sharded-helper"], stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        except Exception:
            helper = None
    dev = plat.device(local)
    rccl_nranks = 0                               # ranks RCCL saw in an actual all-reduce (0: no communicator)
    if world > 1 or rehearse:
        # RCCL prints a version banner on stdout when its first communicator comes up: this process's stdout carries
        # the ONE JSON line, so file descriptor 1 points at stderr while the communicator is created
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
                dist.init_process_group(plat.backend, rank=0, world_size=1, device_id=dev)
                os.environ["CESX_FORCE_COMM_OVERLAP"] = "1"
            elif plat.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(plat.backend)
            one = torch.ones(1, device=dev)
            dist.all_reduce(one)
            rccl_nranks = int(one.item())
            plat.sync()
        finally:
            sys.stdout.flush()
            os.dup2(saved_out, 1)
            os.close(saved_out)

    build_lib, engine, ShardedUpdate = plat.modules()
    if rank == 0:
        build_lib()
    if world > 1:
        dist.barrier()

    p, n, J = args.p, args.n, args.J
    Jg = J * world
    prob = synthetic_problem(p, n)
    eng = engine.Engine(p, n, J, dtype=args.dtype, device=local, J_global=Jg, j_offset=rank * J, seed=1234)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    sh = ShardedUpdate(eng)

    # resident inputs: ring of distinct synthetic batches, G = A U on device
    NB = 4
    gen = plat.generator(dev)
    gen.manual_seed(20240 + rank)
    ustar_d = torch.as_tensor(prob["ustar"], device=dev, dtype=eng.torch_dtype)
    batches = []
    for b in range(NB):
        U = ustar_d + (1.0 + 0.05 * b) * torch.randn((p, J), generator=gen, device=dev, dtype=eng.torch_dtype)
        G = eng.forward_lineal(prob["A"], U)
        batches.append((U, G))
    out = eng.empty(p)
    plat.sync()

    t_hist = [0.0]
    prm0 = engine.step_params(update=args.update)

    # One step = begin (moments, all-reduce, chol: needs nothing from the previous step) + finish
    # (K2 with t_last, K3) + the host's read of the step result, which the driver loop needs for
    # its t_tol test (ces/calibrate.py:387).  The loop is software-pipelined the way
    # ShardedSampler.run is: begin(i+1) is enqueued before result(i) is read, so the host's read
    # overlaps the next Gram instead of idling the GPU.  Every timed step still does all of its
    # work inside the timed region, and every result is read.
    # A launch that carries kernel-bound start / stop events costs its step 20-30 us (time-stamped dispatch), and the
    # sampled steps are inside the timed region: so exactly ONE step of the region -- the middle one -- is sampled, and
    # in it only the DOMINANT kernel's launches carry events (`roofline`, `roofline.profiled_steps`).  The other MFMA
    # kernel and the Gram-end -> K3-start interval are sampled the same way in an untimed step right before the
    # warm-up steps (same process, same warmed-up device, 64 + W steps earlier; `roofline.kernels[*].sampled` says
    # which is which); the rocprofv3 kernel stats under profiles/ are the cross-check for both.
    prof = dict(on=False, steps=0, at=-1, gap_at=-2, mode=True)

    def prof_mode(i):           # the sampled step: start + stop events on K1 and / or K3; another one: the gap's two events only
        return prof["mode"] if i == prof["at"] else 2 if i == prof["gap_at"] else False

    def begin(i):
        U, G = batches[i % NB]
        if prof["on"]:
            eng.profile_enable(prof_mode(i))
            sh.sample_collectives = i == prof["at"]      # (sharded runs: event pairs around the step's all-reduces)
        sh.begin(prm0, U, G, recenter=(i == 0), noise_step=i)

    def finish(i):
        U, G = batches[i % NB]
        if prof["on"]:
            eng.profile_enable(prof_mode(i))
            sh.sample_collectives = i == prof["at"]
            prof["steps"] += int(i == prof["at"])
        prm = engine.step_params(update=args.update, first_step=(i == 0), t_len=min(i, 1), t_last=t_hist[0],
                                 step_index=i)
        sh.finish(prm, U, G, xi=None, out=out)

    stamps = []
    sclk_probe = dict(at=-1, val=None)          # read sysfs sclk at this step of an UNTIMED run: the GPU is busy then

    def run_steps(first, count):
        begin(first)
        res = None
        for i in range(first, first + count):
            finish(i)
            if i + 1 < first + count:
                begin(i + 1)
            if i == sclk_probe["at"]:
                sclk_probe["val"] = sysfs_sclk()
            res = eng.result()
            stamps.append(time.perf_counter())     # host time at which the result of step i was in hand
            t_hist[0] = res.t_new
        return res

    # Untimed pre-warm, time-based.  What it has to remove (measured, tools/ramp_probe.py, profiles/r03_ramp.txt):
    # the step itself is at its steady-state rate from the first 0.25 s of a run -- there is no seconds-long clock
    # ramp -- but whenever the GPU has been IDLE for more than a few milliseconds (it drops into a sleep state:
    # sysfs sclk 95 MHz) the next ~20-30 steps run 10-15 % slower, about 2 ms in all.  A 20-step timed region is
    # 8 ms long, so one such wake-up inside it costs 20 % (0.49 against 0.41 ms/step: the round-1/2 driver numbers,
    # whose timed region started behind a synchronize + gc.collect() + event reads).  So: step continuously for
    # at least CESX_BENCH_PREWARM_S seconds (default 2) in windows of 512 steps until the window time has stopped
    # falling (cap 10 s; every rank takes the same decision), do everything that idles the GPU (garbage collection,
    # event-pool creation, the first time-stamped launches) BEFORE the last pre-warm window, and let nothing but
    # the contract's barrier + synchronize sit between the W warm-up steps and the timed region.
    import gc
    gc.collect()
    gc.disable()                                # no collector pause inside the timed region (a few hundred us each)
    eng.profile_enable(True)                    # creates the event pool outside the timed region
    eng.profile_enable(False)
    prof["on"] = not os.environ.get("CESX_BENCH_NOPROF")
    prewarm_min_s = float(os.environ.get("CESX_BENCH_PREWARM_S", "2.0"))
    prewarm, win_ms, t_pre = 0, [], time.perf_counter()
    WIN = 512
    while True:
        # the first window carries one HIP-event-sampled step (thrown away): whatever the runtime sets up on the
        # first time-stamped launch of a queue happens here
        prof["at"] = 8 if (prewarm == 0 and prof["on"]) else -1
        prof["gap_at"] = 12 if (prewarm == 0 and prof["on"]) else -2
        tb = time.perf_counter()
        run_steps(0, WIN)
        plat.sync()
        win_ms.append((time.perf_counter() - tb) * 1e3 / WIN)
        if prewarm == 0 and prof["on"]:
            eng.profile_read(0), eng.profile_read(1)
            sh.read_collective_ms()
        prewarm += WIN
        spent = time.perf_counter() - t_pre
        # settled: the mean of the last two windows is within 0.5 % of (or above) the two before
        settled = len(win_ms) >= 4 and sum(win_ms[-2:]) >= 0.995 * sum(win_ms[-4:-2])
        done = (spent >= prewarm_min_s and settled) or spent >= 10.0 or prewarm_min_s <= 0
        if world > 1 or rehearse:               # every rank must run the same number of steps (collectives inside)
            flag = torch.tensor([1.0 if done else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item() > 0.5)
        if done:
            break
    prewarm_s = time.perf_counter() - t_pre
    t_hist[0] = 0.0
    # Sharded runs: head + tail (two all-reduces, chol(C) beside the second Gram launch) or the north star's single all-reduce
    # (chol(C) in line behind it)?  One rank says 0.39 against 0.52 ms/step, but one rank moves nothing over xGMI: unless
    # CESX_SINGLE_ALLREDUCE pins it, every rank runs 64 untimed steps in each mode here (the slowest rank's time counts, all
    # ranks see the same two numbers and take the same decision) and the faster mode is the one measured.
    mode_choice = None
    if (world > 1 or rehearse) and os.environ.get("CESX_SINGLE_ALLREDUCE") is None and hasattr(sh, "single_allreduce"):
        cands = {"head_tail": sh, "single": ShardedUpdate(eng, single_allreduce=True)}
        trial = {}
        for name, cand in cands.items():
            sh = cand
            run_steps(0, 32)
            plat.sync()
            if world > 1:
                dist.barrier()
            tb = time.perf_counter()
            run_steps(0, 64)
            plat.sync()
            tt = torch.tensor([(time.perf_counter() - tb) * 1e3 / 64], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            trial[name] = float(tt.item())
        pick = min(trial, key=trial.get)
        sh = cands[pick]
        mode_choice = dict(picked=pick, ms_per_step={k: round(v, 4) for k, v in trial.items()},
                           how="64 untimed steps in each mode behind the pre-warm, max over ranks; CESX_SINGLE_ALLREDUCE=0 / 1 pins one")
        t_hist[0] = 0.0
    # the untimed samples: both MFMA kernels in one step, the gap's two events in another
    pre_k1 = pre_k3 = (0.0, 0)
    pre_gap = None
    if prof["on"]:
        prof["at"], prof["gap_at"], prof["mode"], prof["steps"] = 24, 28, True, 0
        run_steps(0, 48)
        plat.sync()
        pre_gap = eng.profile_gap()
        pre_k1, pre_k3 = eng.profile_read(0), eng.profile_read(1)
        sh.read_collective_ms()
    prof["at"], prof["gap_at"] = -1, -2
    sclk_probe["at"] = 32
    run_steps(0, 64)                            # (the reads above idled the GPU for a moment: back to work first)
    sclk_before, sclk_probe["at"] = sclk_probe["val"], -1
    if args.warmup:
        run_steps(0, args.warmup)
    if world > 1:
        dist.barrier()
    plat.sync()
    # exactly ONE step of the timed region -- the middle one -- is HIP-event sampled, its dominant kernel alone
    dom_is_k3 = pre_k3[0] >= pre_k1[0]
    prof["steps"] = 0
    prof["mode"] = 3 if dom_is_k3 else 4
    prof["at"] = args.warmup + args.steps // 2
    del stamps[:]
    t0 = time.perf_counter()
    res = run_steps(args.warmup, args.steps)
    plat.sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    per_step = np.diff(np.array([t0] + stamps)) * 1e3       # ms between consecutive results (pipelined loop)
    prof["on"] = False
    eng.profile_enable(False)
    # the clock the sampled K3 launch ran at (in-kernel s_memtime / s_memrealtime), a bare-MFMA calibration of this
    # device right behind the timed region, and the sysfs sclk samples around it
    sh.sample_collectives = False
    coll_ms = sh.read_collective_ms()
    gap_ms = pre_gap                                            # end of the second Gram launch -> start of K3 (untimed sample)
    k3_clock = eng.profile_clock() if prof["steps"] else None
    calib_tf, calib_ghz = eng.calibrate_mfma(5.0)
    k3_form = eng.update_form()          # 0 assembled / 1 hk-free / 2 through the Cholesky factor (include/cesx.h)
    sclk_probe["at"] = args.warmup + args.steps + 16
    run_steps(args.warmup + args.steps, 32)     # (untimed: the same steps again, the sysfs clock read while they run)
    plat.sync()
    sclk_after, sclk_probe["at"] = sclk_probe["val"], -1
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if not np.isfinite(res.hk) or not bool(torch.isfinite(out).all()):
        raise SystemExit("non-finite result in the timed region")

    gram_ms, gram_cnt = eng.profile_read(0)
    upd_ms, upd_cnt = eng.profile_read(1)
    where = {"gram_kernel(K1)": "timed region", "update_kernel(K3)": "timed region"}
    if prof["steps"]:                           # the kernel that was not sampled inside the timed region: its untimed sample
        if dom_is_k3:
            gram_ms, gram_cnt = pre_k1[0] * prof["steps"], pre_k1[1] * prof["steps"]
            where["gram_kernel(K1)"] = "untimed step right before the warm-up steps"
        else:
            upd_ms, upd_cnt = pre_k3[0] * prof["steps"], pre_k3[1] * prof["steps"]
            where["update_kernel(K3)"] = "untimed step right before the warm-up steps"
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # algorithmic flops per launch: symmetric-aware Gram of the stacked
    # anomaly (p+n)^2 per particle; fused update GEMM 2 p (2p+n) per particle
    nprof = max(prof["steps"], 1)
    kern = {
        # per step: K1 is two launches (U x U blocks, then the rest beside the Cholesky), K3 one
        # (two for aldi_constant) -- the durations of a step's launches are summed
        "gram_kernel(K1)": dict(ms=gram_ms / nprof, flops=float(p + n) ** 2 * J, launches=gram_cnt / nprof),
        "update_kernel(K3)": dict(ms=upd_ms / nprof, flops=k3_flops_algorithmic(p, n, J, k3_form), launches=upd_cnt / nprof),
    }
    # ... and what the matrix pipe really issues (triangular-aware, block-granular: flops_executed)
    kern["gram_kernel(K1)"]["flops_executed"], kern["update_kernel(K3)"]["flops_executed"] = flops_executed(p, n, J, args.dtype, k3_form)
    for k in kern.values():
        k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
        k["tflops_executed"] = k["flops_executed"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
    dom = max(kern, key=lambda k: kern[k]["ms"])
    peak = MFMA_PEAK_TF[np.dtype(args.dtype).name]
    dname = np.dtype(args.dtype).name
    is_c2 = (p, n, J, dname, args.update) == (256, 256, 65536, "float32", "aldi")
    is_c5 = (p, n, J, dname, args.update) == (512, 512, 32768, "float64", "aldi")
    cfg_name = "C2" if is_c2 else "C5" if is_c5 else "custom"
    # HBM bytes per step of each kernel from the rocprofv3 PMC passes (profiles/traffic.json, taken on the
    # named config with tools/prof_summary.py; collected in separate --pmc runs as the guide prescribes)
    traffic_tab, traffic_stale = {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and cfg_name != "custom":
        try:
            tj = json.load(open(tpath))
            traffic_tab = tj.get(cfg_name, tj if cfg_name == "C2" and "gram_kernel" in tj else {})
        except Exception:
            traffic_tab = {}
        # counters of ANOTHER build of the kernels are not this line's traffic: the table carries the hash of the
        # kernel sources its profile ran from (tools/prof_summary.py), compared with the tree this run uses
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from prof_summary import kernel_sources_sha
            have, want = traffic_tab.get("_src_sha16"), kernel_sources_sha()
            if traffic_tab and have != want:
                traffic_stale = ("profiles/traffic.json[%s] was taken from kernel sources %s (%s), this tree is %s: refused"
                                 % (cfg_name, have, traffic_tab.get("_source", "?"), want))
                traffic_tab = {}
        except Exception as ex:
            traffic_stale, traffic_tab = "traffic.json could not be checked against the kernel sources (%r): refused" % (ex,), {}
    k3name = "update4_kernel" if k3_form == 2 else "update2_kernel" if dname == "float32" else ("update3_kernel" if "update3_kernel" in traffic_tab else "update_kernel")
    tkey = {"gram_kernel(K1)": "gram_kernel", "update_kernel(K3)": k3name}
    traffic = traffic_tab.get(tkey[dom])
    step_s = elapsed / args.steps
    esz = np.dtype(args.dtype).itemsize
    alg_bytes = float(esz * (3 * p + 2 * n)) * J               # s (3p + 2n) per particle-update
    traffic_source = traffic_stale
    if traffic_tab:
        traffic_source = ("profiles/traffic.json[%s] <- profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an "
                          "EARLIER run of this configuration (2 x FETCH + WRITE, the guide's gfx950 correction), not "
                          "counters of this run" % (cfg_name, traffic_tab.get("_source", "?")))
    counter_bytes = (float(sum(v for k, v in traffic_tab.items() if not k.startswith("_") and isinstance(v, (int, float))
                               # (aliases, the set-up's forward map and the first step's recentring pass are not per step)
                               and k not in ("gram_kernel", "update_kernel", "colsum_final_kernel", "rowsum_kernel",
                                             "set_shift_kernel")))
                     if traffic_tab else None)
    mfma_util_tab = traffic_tab.get("_mfma_util", {}) if traffic_tab else {}
    roofline = dict(bound="mfma", kernel=dom, achieved=round(kern[dom]["tflops"], 2), peak=peak,
                    unit="TFLOP/s", frac=round(kern[dom]["tflops"] / peak, 4), traffic=traffic,
                    traffic_source=traffic_source,
                    # `frac` prices the ALGORITHMIC flops of SURVEY.md 8(d) (K3: 2 p (2p + n) J -- the zero blocks of the
                    # triangular sqrt(2hk) L segment included, which the kernel skips); what the hardware did:
                    #   frac_executed  flops the kernel really issues / its time / peak
                    #   mfma_util      SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), PMC passes of an earlier
                    #                  run of this configuration (profiles/traffic.json, same source as `traffic`)
                    frac_executed=round(kern[dom]["tflops_executed"] / peak, 4),
                    mfma_util=mfma_util_tab.get(tkey[dom]),
                    calibration=dict(tflops=round(calib_tf, 2), clock_ghz=round(calib_ghz, 4),
                                     frac_of_calibration=round(kern[dom]["tflops"] / calib_tf, 4) if calib_tf > 0 else None,
                                     what="bare %s loop (operands in registers, 4 accumulators per wave, 2 x 256 threads "
                                          "per CU, random operands), ~5 ms, launched right behind the timed region"
                                          % ("v_mfma_f32_32x32x2_f32" if dname == "float32" else "v_mfma_f64_16x16x4_f64")),
                    avg_launch_ms=round(kern[dom]["ms"], 4), profiled_steps=prof["steps"],
                    kernels={k: dict(avg_launch_ms=round(v["ms"], 4), launches_per_step=v["launches"],
                                     tflops=round(v["tflops"], 2), frac=round(v["tflops"] / peak, 4),
                                     flops=v["flops"], flops_executed=v["flops_executed"],
                                     **({"flops_survey_8d": 2.0 * p * (2 * p + n) * J, "update_form": k3_form,
                                         "flops_note": "K3 through the Cholesky factor: the algorithm needs 2 p (p + 1) + 2 p n flops per "
                                                       "particle (two triangular products + the gain term), not SURVEY 8(d)'s 2 p (2p + n)"}
                                        if k == "update_kernel(K3)" and k3_form == 2 else {}),
                                     frac_executed=round(v["tflops_executed"] / peak, 4),
                                     mfma_util=mfma_util_tab.get(tkey[k]),
                                     sampled=where[k],
                                     traffic=traffic_tab.get(tkey[k]),
                                     hbm_gbs=(round(traffic_tab[tkey[k]] / (v["ms"] * 1e-3) / 1e9, 1)
                                              if traffic_tab.get(tkey[k]) and v["ms"] > 0 else None))
                             for k, v in kern.items()},
                    step_flops_frac=round(sum(v["flops"] for v in kern.values()) / step_s / 1e12 / peak, 4),
                    step_flops_executed_frac=round(sum(v["flops_executed"] for v in kern.values()) / step_s / 1e12 / peak, 4),
                    # the other roofline the north star asks for: HBM bytes (PMC counters) / kernel time / 8 TB/s
                    hbm_gbs=(round(traffic / (kern[dom]["ms"] * 1e-3) / 1e9, 1) if traffic and kern[dom]["ms"] > 0 else None),
                    hbm_frac=(round(traffic / (kern[dom]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                              if traffic and kern[dom]["ms"] > 0 else None),
                    hbm_peak_gbs=HBM_PEAK_GBS,
                    step_algorithmic_bytes=alg_bytes,
                    # every kernel of a step by the same counters (incl. the xi block's write + read and the Gram's
                    # slab round trip): what the step really moves, from the same earlier PMC run as `traffic`
                    step_counter_bytes=counter_bytes,
                    step_algorithmic_hbm_frac=round(alg_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4))
    if world == 1 and not rehearse:
        par = "dp1: one GPU holds the whole ensemble, no collective is issued"
    elif sh.single_allreduce:
        par = ("particle-sharded dp%d over RCCL: ONE all-reduce(sum) of the %d-double fp64 moment buffer per step after "
               "the complete Gram, chol(C) in line" % (world, eng.moments_len()))
    else:
        par = ("particle-sharded dp%d over RCCL: all-reduce(sum) of the %d-double fp64 moment buffer per step, sent in "
               "two pieces (%d-double head on the side stream beside the second Gram launch, then the rest)"
               % (world, eng.moments_len(), eng.moments_uu_len()))
    if world > 1 or rehearse:
        par += ("; mode measured and chosen: %s" % json.dumps(mode_choice["ms_per_step"]) if mode_choice
                else "; mode pinned by CESX_SINGLE_ALLREDUCE=%s" % os.environ.get("CESX_SINGLE_ALLREDUCE"))
    rec = dict(metric="EKS particle-updates/sec", value=Jg * args.steps / elapsed, unit="particle-updates/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * step_s,
               ms_per_step_median=float(np.median(per_step)), ms_per_step_min=float(np.min(per_step)),
               ms_per_step_max=float(np.max(per_step)),
               prewarm_steps=prewarm, prewarm_s=round(prewarm_s, 3), prewarm_window_ms=[round(w, 4) for w in win_ms],
               rccl_nranks=rccl_nranks,
               rccl_comms=(eng.comm_count() if hasattr(eng, "comm_count") else 0),      # 2: one communicator per stream (include/cesx.h)
               clock=dict(k3_ghz=round(k3_clock, 4) if k3_clock else None,
                          k3_how="s_memtime / s_memrealtime of one wave of the last HIP-event-sampled update launch (%s)"
                                 % where["update_kernel(K3)"],
                          calibration_ghz=round(calib_ghz, 4), sclk_before=sclk_before, sclk_after=sclk_after,
                          sclk_how="sysfs pp_dpm_sclk level marked current, read by the host WHILE an untimed run of the "
                                   "same steps is in flight (an idle GPU reads 100-700 MHz): 32 steps before the warm-up "
                                   "steps, and 16 steps into a 32-step run behind the timed region and the calibration loop"),
               higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype={"float32": "f32", "float64": "f64"}[dname], data="synthetic",
               config=dict(workload="%s per GPU: synthetic linear-Gaussian forward map, J=%d particles/GPU "
                                    "(J_global=%d), d=p=%d, n_obs=%d, update=%s, default Frobenius time step, "
                                    "on-device Philox noise" % (cfg_name, J, Jg, p, n, args.update),
                           J_per_gpu=J, J_global=Jg, p=p, n_obs=n, update=args.update, parallelism=par,
                           timed_region="the update engine on HBM-resident inputs: K1 moments, (all-reduce), K2, K3 "
                                        "with on-device noise and the host's read of hk / t / metrics of every step. "
                                        "A ring of 4 resident (U, G = A U) batches is cycled; the forward map and the "
                                        "feedback of U_next are NOT in the timed step (see e2e.device_chain for the "
                                        "chained loop); %d untimed pre-warm steps (%.1f s) precede the warm-up" % (prewarm, prewarm_s)),
               roofline=roofline,
               # what a SCALE line needs to explain its own efficiency (rank 0, the sampled step of the timed region)
               sampled_step=dict(gram_end_to_k3_start_ms=round(gap_ms, 4) if gap_ms is not None else None,
                                 collective_mode=("none" if world == 1 and not rehearse else
                                                  "single" if sh.single_allreduce else "head+tail"),
                                 collectives_per_step=(0 if world == 1 and not rehearse else 1 if sh.single_allreduce else 2),
                                 mode_choice=mode_choice,
                                 collective_ms={k: dict(doubles=v[0], ms=round(v[1], 4)) for k, v in coll_ms.items()},
                                 how="the interval: HIP events of an UNTIMED step of their own right before the warm-up steps "
                                     "(nothing else time-stamped in it): kernel-bound stop of the second Gram launch -> "
                                     "kernel-bound start of K3; the collectives: a recorded event pair around each all-reduce of "
                                     "the timed region's sampled step, on the stream that issues it (the pair itself adds a few "
                                     "us to that step)"))
    if world == 1 and not rehearse and not args.no_extras:
        del batches, out, sh, eng
        torch.cuda.empty_cache()
        rec["parity_err"] = parity_check(engine, prob, p, n, args.dtype, args.update)
        if is_c2:
            # the other single-GPU configurations of BASELINE.json, short legs (never `value`):
            #   C5      configs[4] per GPU: fp64, p = n_obs = 512, J = 32 768 (the reference's own precision)
            #   C2_f64  the headline shape at the reference's precision
            #   C4      configs[3]: p = 64, n_obs = 50, J = 8 192 -- the update alone (latency-bound regime) and, at a J the
            #           host finishes in seconds, end to end with the host Darcy map
            extra = {}
            extra["C5"] = engine_leg(engine, "C5 per GPU", 512, 512, 32768, "float64", 10, local, prewarm_s=0.5)
            extra["C2_f64"] = engine_leg(engine, "C2 shape in fp64", 256, 256, 65536, "float64", 10, local, prewarm_s=0.5)
            extra["C4"] = engine_leg(engine, "C4 update only", 64, 50, 8192, "float32", 40, local, prewarm_s=0.4)
            try:
                extra["C4"]["e2e_darcy"] = darcy_leg(engine, local)
            except Exception as ex:             # (scipy missing, ...): the engine legs above still stand
                extra["C4"]["e2e_darcy"] = dict(error=repr(ex))
            try:
                extra["small_J"] = small_j_leg(engine, local)
            except Exception as ex:
                extra["small_J"] = dict(error=repr(ex))
            # every other rule / time step / dense problem of the same reference function at the headline shape
            extra["variants"] = variants_leg(engine, p, n, J, args.dtype, local)
            # the sharded path on this one GPU, RCCL really initialised and issued, both collective modes (fresh processes;
            # this one stays idle meanwhile).  NO 1 -> 8 scaling curve is measured by this line: one rank only.
            if helper is not None:
                try:
                    torch.cuda.synchronize()
                    txt, _ = helper.communicate("go\n", timeout=600)
                    extra["sharded_one_rank"] = json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])
                    extra["sharded_one_rank"]["note"] = ("one rank: the collectives are really issued through RCCL but move nothing "
                                                         "over xGMI; no scaling curve is implied")
                except Exception as ex:
                    extra["sharded_one_rank"] = dict(error=repr(ex))
                    try:
                        helper.kill()
                    except Exception:
                        pass
                helper = None
            rec["extra"] = extra
        rec["e2e"] = e2e_block(engine, prob, p, n, J, args.dtype, args.update, local)
    if world == 1 and not args.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(prob, p, n, J, np.dtype(args.dtype).type)
    if helper is not None:                       # (not asked: it exits on the empty line)
        try:
            helper.communicate("\n", timeout=10)
        except Exception:
            helper.kill()
    print(json.dumps(rec), flush=True)
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    if "--sharded-helper" in sys.argv:
        sharded_helper_main()
    else:
        main()
