#!/usr/bin/env python3
"""dev tool: how long after start-up does the C2 step reach its steady state?

Steps the engine exactly as bench.py does (pipelined begin / finish / result over a ring of 4 batches) for
``--seconds`` of wall time and prints, per 0.25-s window: steps, mean / median ms per step, one HIP-event
sample of K1 and K3 (kernel-bound events), and the sclk lines of sysfs ``pp_dpm_sclk`` that carry a ``*``.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def sclk():
    out = []
    for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            cur = [ln.strip() for ln in open(path) if "*" in ln]
            out.append(cur[0] if cur else "?")
        except OSError:
            pass
    return out[:1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--window", type=float, default=0.25)
    ap.add_argument("--idle", type=float, default=0.0, help="sleep this long (GPU idle) half way through")
    ap.add_argument("--noprof", action="store_true")
    ap.add_argument("--gaps", action="store_true", help="after the run: 20-step timed regions behind idle gaps of 0 .. 200 ms")
    args = ap.parse_args()
    from ces_amd import engine
    from ces_amd.dist import ShardedUpdate
    p = n = 256
    J = 65536
    dev = torch.device("cuda", 0)
    prob = bench.synthetic_problem(p, n)
    eng = engine.Engine(p, n, J, dtype="float32", device=0, seed=1234)
    eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
    sh = ShardedUpdate(eng)
    gen = torch.Generator(device=dev)
    gen.manual_seed(20240)
    ustar_d = torch.as_tensor(prob["ustar"], device=dev, dtype=eng.torch_dtype)
    batches = []
    for b in range(4):
        U = ustar_d + (1.0 + 0.05 * b) * torch.randn((p, J), generator=gen, device=dev, dtype=eng.torch_dtype)
        batches.append((U, eng.forward_lineal(prob["A"], U)))
    out = eng.empty(p)
    torch.cuda.synchronize()
    prm0 = engine.step_params(update="aldi")
    t_hist = [0.0]
    eng.profile_enable(True)
    eng.profile_enable(False)

    def begin(i, prof=False):
        U, G = batches[i % 4]
        eng.profile_enable(prof)
        sh.begin(prm0, U, G, recenter=(i == 0), noise_step=i)

    def finish(i, prof=False):
        U, G = batches[i % 4]
        eng.profile_enable(prof)
        prm = engine.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_hist[0], step_index=i)
        sh.finish(prm, U, G, xi=None, out=out)

    rows = []
    t_start = time.perf_counter()
    i = 0
    begin(0)
    idled = args.idle <= 0
    while time.perf_counter() - t_start < args.seconds:
        w0 = time.perf_counter()
        stamps = [w0]
        first = i
        sample_at = -1 if args.noprof else i + 8
        while time.perf_counter() - w0 < args.window:
            finish(i, prof=(i == sample_at))
            begin(i + 1, prof=(i + 1 == sample_at))
            res = eng.result()
            t_hist[0] = res.t_new if i % 4000 else 0.0        # keep t bounded over a long run
            stamps.append(time.perf_counter())
            i += 1
        eng.profile_enable(False)
        k1 = k3 = None
        if not args.noprof:
            torch.cuda.synchronize()
            k1, c1 = eng.profile_read(0)
            k3, c3 = eng.profile_read(1)
        d = np.diff(np.array(stamps)) * 1e3
        rows.append(dict(t=round(w0 - t_start, 3), steps=i - first, mean_ms=round(float(d.mean()), 4),
                         med_ms=round(float(np.median(d)), 4), k1_ms=k1 and round(k1, 4), k3_ms=k3 and round(k3, 4),
                         sclk=sclk()))
        print(json.dumps(rows[-1]), flush=True)
        if not idled and time.perf_counter() - t_start > args.seconds / 2:
            torch.cuda.synchronize()
            time.sleep(args.idle)
            idled = True
            print(json.dumps(dict(idle_s=args.idle)), flush=True)
    if args.gaps:
        eng.profile_enable(False)
        torch.cuda.synchronize()

        def run_steps(first, count):                 # bench.py's loop: nothing enqueued beyond the last step
            begin(first)
            for k in range(first, first + count):
                finish(k)
                if k + 1 < first + count:
                    begin(k + 1)
                res = eng.result()
                t_hist[0] = res.t_new if k % 4000 else 0.0
            return first + count
        for gap in (0.0, 0.0002, 0.001, 0.003, 0.01, 0.03, 0.1, 0.3, 0.0):
            vals = []
            for rep in range(6):
                i = run_steps(i, 200)                 # the GPU is busy and warm, then the gap, then the timed 20
                torch.cuda.synchronize()
                if gap:
                    time.sleep(gap)
                t0 = time.perf_counter()
                i = run_steps(i, 20)
                torch.cuda.synchronize()
                vals.append((time.perf_counter() - t0) / 20 * 1e3)
            print(json.dumps(dict(gap_ms=gap * 1e3, ms_per_step=[round(v, 4) for v in vals])), flush=True)


if __name__ == "__main__":
    main()
