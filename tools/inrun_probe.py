"""dev tool: the `time_step='constant'` in-a-run leg several times with host timers around the calls of a step
(a leg sometimes runs at twice the time with the GPU waiting ~60 us for every launch: which host call is slow?)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ces_amd import engine, dist
from ces_amd.dist import ShardedSampler
from ces_amd.utils import lineal
p = n = 256; J = 65536
prob = bench.synthetic_problem(p, n)
rng = np.random.default_rng(3)
U0 = prob["ustar"] + rng.standard_normal((p, J))
model = lineal(prob["A"])
os.environ["CESX_LINEAL_FAST"] = "0"
kw = dict(update="aldi", time_step="constant", delta_t=0.0304)
acc = {}
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(cls, name, g)
for nm in ("begin", "finish", "result"): wrap(dist.ShardedUpdate, nm)
wrap(ShardedSampler, "_forward")
def cpu_stat():
    try:
        return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(),
      "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None, flush=True)
def legs(tag, nrep):
    for rep in range(nrep):
        st0 = cpu_stat()
        eng = engine.Engine(p, n, J, dtype="float32", device=0, seed=77)
        for steps in (8, 120):
            smp = ShardedSampler(eng, p, n, J); smp.T = steps
            acc.clear()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            smp.run(prob["y"], U0, model, prob["Gamma"], prob["mu"], prob["sigma"], prob["ustar"], t_tol=1e30, **kw)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        st1 = cpu_stat()
        print(tag, rep, "%.4f ms/step" % (1e3 * el / 120), " host us/step:", {k: round(1e6 * v / 120, 1) for k, v in acc.items()},
              "throttled %d of %d periods, %d us" % (st1.get("nr_throttled", 0) - st0.get("nr_throttled", 0), st1.get("nr_periods", 0) - st0.get("nr_periods", 0),
                                                    st1.get("throttled_usec", 0) - st0.get("throttled_usec", 0)), flush=True)
        del eng, smp
legs("default-stream", int(sys.argv[1]) if len(sys.argv) > 1 else 8)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    legs("own-stream", int(sys.argv[1]) if len(sys.argv) > 1 else 8)
