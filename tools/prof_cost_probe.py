#!/usr/bin/env python3
"""dev tool: what one HIP-event-sampled step costs the pipelined loop, per cesx_profile_enable mode (0 off, 1 both MFMA
kernels, 2 the gap's two events, 3 the update launch alone, 4 the moments launches alone): ms of the steps around the
sampled one and the host time of its finish / begin calls.  Round 3: mode 3 +30 us spread over three steps, modes 1 and 4
+100 us (the events on the Gram launches delay the hand-over to the side stream), mode 2 nothing measurable."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd import engine
from ces_amd.dist import ShardedUpdate
p = n = 256; J = 65536
prob = bench.synthetic_problem(p, n)
eng = engine.Engine(p, n, J, dtype="float32", device=0, seed=1234)
eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
sh = ShardedUpdate(eng)
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
us = torch.as_tensor(prob["ustar"], device=dev, dtype=torch.float32)
batches = []
for b in range(4):
    U = us + torch.randn((p, J), generator=gen, device=dev, dtype=torch.float32)
    batches.append((U, eng.forward_lineal(prob["A"], U)))
out = eng.empty(p); prm0 = engine.step_params(update="aldi"); t_last = [0.0]
eng.profile_enable(True); eng.profile_enable(False)
def begin(i, m=0):
    U, G = batches[i % 4]; eng.profile_enable(m); sh.begin(prm0, U, G, recenter=(i == 0), noise_step=i)
def finish(i, m=0):
    U, G = batches[i % 4]; eng.profile_enable(m)
    prm = engine.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_last[0], step_index=i)
    sh.finish(prm, U, G, xi=None, out=out)
def run(first, count, at=-1, mode=0):
    stamps, hostb, hostf = [], [], []
    begin(first)
    for i in range(first, first + count):
        t0 = time.perf_counter(); finish(i, mode if i == at else 0); t1 = time.perf_counter()
        if i + 1 < first + count: begin(i + 1, mode if i + 1 == at else 0)
        t2 = time.perf_counter()
        res = eng.result(); t_last[0] = res.t_new if i % 4000 else 0.0
        stamps.append(time.perf_counter()); hostf.append(t1 - t0); hostb.append(t2 - t1)
    return np.diff(np.array(stamps)) * 1e3, np.array(hostf) * 1e6, np.array(hostb) * 1e6
run(0, 3000)
for mode in (0, 3, 4, 1, 2):
    d, hf, hb = run(0, 40, at=20, mode=mode)
    torch.cuda.synchronize(); eng.profile_gap(); eng.profile_read(0); eng.profile_read(1)
    print("mode", mode, "step ms around the sampled one:", np.round(d[16:24], 4), "host us finish/begin at sampled:", round(hf[20], 1), round(hb[19], 1), "typical:", round(np.median(hf), 1), round(np.median(hb), 1))
    run(0, 200)
