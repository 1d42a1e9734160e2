#!/usr/bin/env python3
"""dev tool: A/B of engine build-time switches (environment variables read by cesx_create) in ONE process.

    python tools/ab_env.py "CESX_POLL_JOIN=0" "" "CESX_NOISE_LOOKAHEAD=0,CESX_HKFREE=0" [--rounds 3 --steps 400]

One engine per configuration (the variables are set only while that engine is created), the C2 step driven as in
bench.py (pipelined begin / finish / result over a ring of 4 resident batches), the configurations interleaved
round by round so that clock drift hits all of them alike.  Prints per configuration: ms/step of every round and
the Gram-end -> K3-start gap of gap-only sampled steps, K1 / K3 by kernel-bound events.  At most four configurations:
HIP multiplexes the streams of a priority level onto four hardware queues, a fifth engine's side stream shares one.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Leg:
    def __init__(self, spec, p, n, J, dtype, prob, batches):
        from ces_amd import engine
        from ces_amd.dist import ShardedUpdate
        self.engine = engine
        self.spec = spec
        env = dict(kv.split("=", 1) for kv in spec.split(",") if kv)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            self.eng = engine.Engine(p, n, J, dtype=dtype, device=0, seed=1234)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self.eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
        self.sh = ShardedUpdate(self.eng)
        self.batches = batches
        self.out = self.eng.empty(p)
        self.t_last = 0.0
        self.i = 0
        self.ms = []
        self.gaps = []
        self.k1, self.k3 = [], []
        self.prm0 = engine.step_params(update="aldi")

    def begin(self, i, prof=0):
        U, G = self.batches[i % 4]
        self.eng.profile_enable(prof)
        self.sh.begin(self.prm0, U, G, recenter=(i == 0), noise_step=i)

    def finish(self, i, prof=0):
        U, G = self.batches[i % 4]
        self.eng.profile_enable(prof)
        prm = self.engine.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=self.t_last, step_index=i)
        self.sh.finish(prm, U, G, xi=None, out=self.out)

    def run(self, count, sample=False, kernels=False):
        first = self.i
        gap_at = first + count // 2 if (sample or kernels) else -1
        mode = 1 if kernels else 2
        self.begin(first)
        for k in range(first, first + count):
            self.finish(k, prof=mode if k == gap_at else 0)
            if k + 1 < first + count:
                self.begin(k + 1, prof=mode if k + 1 == gap_at else 0)
            res = self.eng.result()
            self.t_last = res.t_new if k % 4000 else 0.0
        self.i = first + count
        self.eng.profile_enable(False)
        if kernels:
            torch.cuda.synchronize()
            k1, c1 = self.eng.profile_read(0)
            k3, c3 = self.eng.profile_read(1)
            self.k1.append(round(k1 * 1e3, 1))
            self.k3.append(round(k3 * 1e3, 1))
        if sample:
            torch.cuda.synchronize()
            g = self.eng.profile_gap()
            self.eng.profile_read(0)
            self.eng.profile_read(1)
            if g is not None and g > 0:
                self.gaps.append(g * 1e3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="+", help='comma-separated VAR=VALUE lists, "" = defaults')
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--J", type=int, default=65536)
    ap.add_argument("--p", type=int, default=256)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--dtype", default="float32")
    args = ap.parse_args()
    from ces_amd import engine
    p, n, J = args.p, args.n, args.J
    dev = torch.device("cuda", 0)
    prob = bench.synthetic_problem(p, n)
    tdt = torch.float32 if args.dtype == "float32" else torch.float64
    gen = torch.Generator(device=dev)
    gen.manual_seed(20240)
    ustar_d = torch.as_tensor(prob["ustar"], device=dev, dtype=tdt)
    # (HIP multiplexes the streams of one priority level onto four hardware queues: a fifth engine's side stream shares
    #  one and its step reads 0.7 ms -- at most four engines alive at a time)
    if len(args.specs) > 4:
        sys.exit("at most 4 configurations per process (one hardware queue per side stream)")
    e0 = engine.Engine(p, n, J, dtype=args.dtype, device=0, seed=1234)
    batches = []
    for b in range(4):
        U = ustar_d + (1.0 + 0.05 * b) * torch.randn((p, J), generator=gen, device=dev, dtype=tdt)
        batches.append((U, e0.forward_lineal(prob["A"], U)))
    torch.cuda.synchronize()
    del e0
    legs = [Leg(s, p, n, J, args.dtype, prob, batches) for s in args.specs]
    for leg in legs:              # warm every engine (first-step recentring, noise buffers, lazy allocations)
        leg.run(64)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        legs[0].run(256)
    for r in range(args.rounds):
        for leg in legs:
            leg.run(64)
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            leg.run(args.steps)
            torch.cuda.synchronize()
            leg.ms.append((time.perf_counter() - w0) / args.steps * 1e3)
            for _ in range(4):
                leg.run(24, sample=True)
            leg.run(24, kernels=True)
    for leg in legs:
        print(json.dumps(dict(spec=leg.spec or "(defaults)", ms_per_step=[round(v, 4) for v in leg.ms],
                              best=round(min(leg.ms), 4), gap_us=[round(g, 1) for g in leg.gaps],
                              gap_med=round(float(np.median(leg.gaps)), 1) if leg.gaps else None,
                              k1_us=leg.k1, k3_us=leg.k3)), flush=True)


if __name__ == "__main__":
    main()
