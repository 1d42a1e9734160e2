cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out &&
python -m pytest tests -x -q -m gpu > gpurun_out/r5_gputests16.txt 2>&1; tail -3 gpurun_out/r5_gputests16.txt;
python3 - <<'PY' > gpurun_out/c4_f64.txt 2>&1
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ces_amd import engine
for rnd in range(2):
  for sm in ("0", "1"):
    os.environ["CESX_UPDATE_SMALL"] = sm
    for dt in ("float32", "float64"):
        r = bench.engine_leg(engine, "C4 update only", 64, 50, 8192, dt, 200, 0, prewarm_s=0.4)
        k = r["roofline"]["kernels"]
        print("small=%s %s: C4 %.4f ms/step  K1 %.4f K3 %.4f" % (sm, dt, r["ms_per_step"], k["gram_kernel(K1)"]["avg_launch_ms"], k["update_kernel(K3)"]["avg_launch_ms"]), flush=True)
PY
cat gpurun_out/c4_f64.txt
