#!/usr/bin/env python3
"""dev tool: the chained device loop (bench.py e2e.device_chain / _full_gram) under engine switches, one process per leg:
    python tools/chain_ab.py "" "CESX_FUSE_CENTER=1" """
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import os, sys, json, numpy as np, torch
sys.path.insert(0, %r)
import bench
from ces_amd import engine
prob = bench.synthetic_problem(256, 256)
r = bench.e2e_block(engine, prob, 256, 256, 65536, 'float32', 'aldi', 0) if False else None
from ces_amd.dist import ShardedSampler
from ces_amd.utils import lineal
import time
rng = np.random.default_rng(3)
U0 = prob['ustar'] + rng.standard_normal((256, 65536))
model = lineal(prob['A'])
eng = engine.Engine(256, 256, 65536, dtype='float32', device=0, seed=77)
out = {}
for T, timed, fast in ((8, False, True), (300, True, True), (8, False, False), (300, True, False)):
    os.environ['CESX_LINEAL_FAST'] = '1' if fast else '0'
    smp = ShardedSampler(eng, 256, 256, 65536); smp.T = T
    torch.cuda.synchronize(); t0 = time.perf_counter()
    smp.run(prob['y'], U0, model, prob['Gamma'], prob['mu'], prob['sigma'], prob['ustar'], update='aldi', t_tol=1e30)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    if timed: out['fast' if fast else 'full'] = round(1e3 * el / T, 4)
print(json.dumps(out))
""" % ROOT
for spec in sys.argv[1:] or [""]:
    env = dict(os.environ)
    env.update(dict(kv.split("=", 1) for kv in spec.split(",") if kv))
    for rep in range(2):
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(repr(spec), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
