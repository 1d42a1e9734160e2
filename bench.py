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
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_PEAK_TF = {"float32": 157.3, "float64": 78.6}   # dense matrix peaks, f32-in / f64 MFMA


def synthetic_problem(p, n, seed=20240):
    """SURVEY.md 8(d) [decision]: A ~ N(0,1)/sqrt(p), Gamma = 0.01 I, mu = 0, Sigma = 100 I."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, p)) / np.sqrt(p)
    ustar = rng.standard_normal((p, 1))
    Gamma = 0.01 * np.eye(n)
    y = (A @ ustar).ravel() + 0.1 * rng.standard_normal(n)
    return dict(A=A, ustar=ustar, Gamma=Gamma, y=y, mu=np.zeros((p, 1)), sigma=100.0 * np.eye(p))


def cpu_baseline(prob, p, n, J, dtype, budget_s=12.0, max_steps=12):
    """The numpy CPU path (oracle, factored form = the only form that fits in
    host memory at J = 65 536) on this box's host cores."""
    from oracle import ces_numpy as oc
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    rng = np.random.default_rng(1)
    U = (prob["ustar"] + rng.standard_normal((p, J))).astype(dtype)
    G = (prob["A"].astype(dtype) @ U)
    st = oc.OracleState(p, n, J, prob["mu"], prob["sigma"], prob["ustar"])
    steps, t0 = 0, time.perf_counter()
    while steps < max_steps and (steps == 0 or time.perf_counter() - t0 < budget_s):
        xi = np.random.normal(0, 1, [p, J])              # ces/calibrate.py:488 draws inside the update
        st.trace_len = 1 if steps == 0 else 2
        oc.factored_step(st, prob["y"], U, G, prob["Gamma"], xi, update="aldi", dtype=dtype)
        steps += 1
    el = time.perf_counter() - t0
    return dict(value=J * steps / el, unit="particle-updates/s", cores=int(cores), kind="port",
                sample="%d steps of oracle.factored_step (J x J-free numpy restatement of "
                       "ces/calibrate.py:451-490, %s, incl. np.random.normal) at J=%d, p=%d, n_obs=%d; "
                       "%.1f s" % (steps, np.dtype(dtype).name, J, p, n, el))


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rehearse = world == 1 and os.environ.get("CESX_FORCE_COLLECTIVES") == "1"   # one-rank run of the N > 1 code path
    if world > 1 or rehearse:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            os.environ["CESX_FORCE_COMM_OVERLAP"] = "1"
        else:
            dist.init_process_group("nccl", device_id=dev)

    from ces_amd import build, engine
    from ces_amd.dist import ShardedUpdate
    if rank == 0:
        build.build_lib()
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
    gen = torch.Generator(device=dev)
    gen.manual_seed(20240 + rank)
    ustar_d = torch.as_tensor(prob["ustar"], device=dev, dtype=eng.torch_dtype)
    batches = []
    for b in range(NB):
        U = ustar_d + (1.0 + 0.05 * b) * torch.randn((p, J), generator=gen, device=dev, dtype=eng.torch_dtype)
        G = eng.forward_lineal(prob["A"], U)
        batches.append((U, G))
    out = eng.empty(p)
    torch.cuda.synchronize()

    t_hist = [0.0]
    prm0 = engine.step_params(update=args.update)

    # One step = begin (moments, all-reduce, chol: needs nothing from the previous step) + finish
    # (K2 with t_last, K3) + the host's read of the step result, which the driver loop needs for
    # its t_tol test (ces/calibrate.py:387).  The loop is software-pipelined the way
    # ShardedSampler.run is: begin(i+1) is enqueued before result(i) is read, so the host's read
    # overlaps the next Gram instead of idling the GPU.  Every timed step still does all of its
    # work inside the timed region, and every result is read.
    # HIP events around K1 / K3 cost ~4 us each (a barrier packet per record), so the live
    # per-kernel durations are taken on every PROF_EVERY-th step of the timed region
    PROF_EVERY = 4
    prof = dict(on=False, steps=0)

    def begin(i):
        U, G = batches[i % NB]
        if prof["on"]:
            eng.profile_enable(i % PROF_EVERY == 0)
        sh.begin(prm0, U, G, recenter=(i == 0))

    def finish(i):
        U, G = batches[i % NB]
        if prof["on"]:
            eng.profile_enable(i % PROF_EVERY == 0)
            prof["steps"] += int(i % PROF_EVERY == 0)
        prm = engine.step_params(update=args.update, first_step=(i == 0), t_len=min(i, 1), t_last=t_hist[0],
                                 step_index=i)
        sh.finish(prm, U, G, xi=None, out=out)

    def run_steps(first, count):
        begin(first)
        res = None
        for i in range(first, first + count):
            finish(i)
            if i + 1 < first + count:
                begin(i + 1)
            res = eng.result()
            t_hist[0] = res.t_new
        return res

    # untimed pre-warm before the W warmup steps: the first ~30 steps after start-up run 3-5 % slower
    # (clock / power state, code objects, allocator); 64 steps = 32 ms reach the steady state
    run_steps(0, int(os.environ.get("CESX_BENCH_PREWARM", "64")))
    t_hist[0] = 0.0
    if args.warmup:
        run_steps(0, args.warmup)
    eng.profile_enable(True)                    # creates the event pool outside the timed region
    eng.profile_read(0), eng.profile_read(1)
    eng.profile_enable(False)
    prof["on"] = not os.environ.get("CESX_BENCH_NOPROF")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_steps(args.warmup, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof["on"] = False
    eng.profile_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if not np.isfinite(res.hk) or not bool(torch.isfinite(out).all()):
        raise SystemExit("non-finite result in the timed region")

    gram_ms, gram_cnt = eng.profile_read(0)
    upd_ms, upd_cnt = eng.profile_read(1)
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
        "update_kernel(K3)": dict(ms=upd_ms / nprof, flops=2.0 * p * (2 * p + n) * J, launches=upd_cnt / nprof),
    }
    for k in kern.values():
        k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
    dom = max(kern, key=lambda k: kern[k]["ms"])
    peak = MFMA_PEAK_TF[np.dtype(args.dtype).name]
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    is_c2 = (p, n, J, np.dtype(args.dtype).name, args.update) == (256, 256, 65536, "float32", "aldi")
    if os.path.exists(tpath) and is_c2:          # the PMC passes were taken on C2
        try:
            traffic = json.load(open(tpath)).get(dom.split("(")[0])
        except Exception:
            traffic = None
    roofline = dict(bound="mfma", kernel=dom, achieved=round(kern[dom]["tflops"], 2), peak=peak,
                    unit="TFLOP/s", frac=round(kern[dom]["tflops"] / peak, 4), traffic=traffic,
                    avg_launch_ms=round(kern[dom]["ms"], 4), profiled_steps=prof["steps"],
                    kernels={k: dict(avg_launch_ms=round(v["ms"], 4), launches_per_step=v["launches"],
                                     tflops=round(v["tflops"], 2), frac=round(v["tflops"] / peak, 4))
                             for k, v in kern.items()},
                    step_flops_frac=round(sum(v["flops"] for v in kern.values()) /
                                          (elapsed / args.steps) / 1e12 / peak, 4))
    rec = dict(metric="EKS particle-updates/sec", value=Jg * args.steps / elapsed, unit="particle-updates/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps,
               higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype={"float32": "f32", "float64": "f64"}[np.dtype(args.dtype).name], data="synthetic",
               config=dict(workload="%s per GPU: synthetic linear-Gaussian forward map, J=%d particles/GPU "
                                    "(J_global=%d), d=p=%d, n_obs=%d, update=%s, default Frobenius time step, "
                                    "on-device Philox noise"
                                    % ("C2" if is_c2 else "C5" if (p, n, J, args.dtype) == (512, 512, 32768, "float64")
                                       else "custom", J, Jg, p, n, args.update),
                           J_per_gpu=J, J_global=Jg, p=p, n_obs=n, update=args.update,
                           parallelism="particle-sharded dp%d, all-reduce(sum) of the %d-double fp64 moment buffer "
                                       "per step in two pieces (%d-double head beside the second Gram launch)"
                                       % (world, eng.moments_len(), eng.moments_uu_len())),
               roofline=roofline)
    if world == 1 and not args.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(prob, p, n, J, np.dtype(args.dtype).type)
    print(json.dumps(rec), flush=True)
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
