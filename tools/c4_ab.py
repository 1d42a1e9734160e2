import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ces_amd import engine
for fv in ("0", "1"):
    os.environ["CESX_FUSE_CENTER"] = fv
    r = bench.engine_leg(engine, "C4 update only", 64, 50, 8192, "float32", 200, 0, prewarm_s=0.4)
    print("CESX_FUSE_CENTER=%s: C4 %.4f ms/step" % (fv, r["ms_per_step"]), flush=True)
del os.environ["CESX_FUSE_CENTER"]
r = bench.engine_leg(engine, "C4 update only", 64, 50, 8192, "float32", 200, 0, prewarm_s=0.4)
print("default: C4 %.4f ms/step" % r["ms_per_step"], flush=True)
print(json.dumps(bench.small_j_leg(engine, 0), indent=1))
