import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ces_amd import engine
for spec in ("", "CESX_NO_NOISE_PREFETCH=1", "CESX_NOISE_LOOKAHEAD=0"):
    for kv in spec.split(","):
        if kv: k, v = kv.split("="); os.environ[k] = v
    r = bench.engine_leg(engine, "C4 update only", 64, 50, 8192, "float32", 300, 0, prewarm_s=0.4)
    r2 = bench.engine_leg(engine, "p256 small J", 256, 50, 2048, "float32", 300, 0, prewarm_s=0.4)
    print("%-28s C4 %.4f ms/step   (256,50,2048) %.4f" % (spec, r["ms_per_step"], r2["ms_per_step"]), flush=True)
    for kv in spec.split(","):
        if kv: os.environ.pop(kv.split("=")[0])
