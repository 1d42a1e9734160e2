"""dev tool: the small-shape legs under an engine switch.   python tools/c4_ab.py ENVVAR [values...]"""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ces_amd import engine
var = sys.argv[1] if len(sys.argv) > 1 else "CESX_UPDATE_SMALL"
vals = sys.argv[2:] or ["0", "1"]
for rnd in range(2):
    for fv in vals:
        os.environ[var] = fv
        r = bench.engine_leg(engine, "C4 update only", 64, 50, 8192, "float32", 200, 0, prewarm_s=0.4)
        k = r["roofline"]["kernels"]
        print("%s=%s: C4 %.4f ms/step  K1 %.4f K3 %.4f" % (var, fv, r["ms_per_step"], k["gram_kernel(K1)"]["avg_launch_ms"],
                                                           k["update_kernel(K3)"]["avg_launch_ms"]), flush=True)
for fv in vals:
    os.environ[var] = fv
    sj = bench.small_j_leg(engine, 0)
    print(var, fv, json.dumps(sj)[:1500], flush=True)
