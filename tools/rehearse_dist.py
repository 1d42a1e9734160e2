#!/usr/bin/env python3
"""dev tool: one-rank rehearsal of the sharded step with host-side timing of every call
(CESX_FORCE_COLLECTIVES=1 python tools/rehearse_dist.py)."""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ["CESX_FORCE_COLLECTIVES"] = "1"; os.environ["CESX_FORCE_COMM_OVERLAP"] = "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from ces_amd import engine
from ces_amd.dist import ShardedUpdate
import bench
p = n = 256; J = 65536
prob = bench.synthetic_problem(p, n)
eng = engine.Engine(p, n, J, dtype="float32", device=0, J_global=J, j_offset=0, seed=1)
eng.set_problem(prob["y"], prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"])
sh = ShardedUpdate(eng)
U = torch.randn((p, J), device="cuda"); G = eng.forward_lineal(prob["A"], U); out = eng.empty(p)
prm0 = engine.step_params(update="aldi")
T = {}
def tm(name, f, *a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
def begin(i):
    nuu = eng.moments_uu_len()
    mom = tm("moments_uu", eng.moments_uu, U, G)
    cur = torch.cuda.current_stream(); cs = sh._cs or eng.side_stream(); sh._cs = cs
    tm("wait_stream", cs.wait_stream, cur)
    with torch.cuda.stream(cs):
        tm("allreduce_head", dist.all_reduce, mom[:nuu])
        tm("chol_async", eng.chol_async, prm0, mom)
    tm("moments_rest", eng.moments_rest, U, G, mom)
    tm("allreduce_rest", dist.all_reduce, mom[nuu:])
    sh._mom = mom
t_last = 0.0
sh.recenter(U, G)
for i in range(30):
    begin(i)
    prm = engine.step_params(update="aldi", first_step=(i == 0), t_len=min(i, 1), t_last=t_last, step_index=i)
    tm("finish", sh.finish, prm, U, G, None, out)
    res = tm("result", eng.result); t_last = res.t_new
for k, v in T.items():
    print("%-16s mean %7.1f us  (last 20: %7.1f)" % (k, 1e6 * np.mean(v), 1e6 * np.mean(v[-20:])))
dist.destroy_process_group()
