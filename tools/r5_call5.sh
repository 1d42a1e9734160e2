#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5_gputests5.txt 2>&1 || { grep -E "^FAILED|^ERROR" gpurun_out/r5_gputests5.txt; }
tail -3 gpurun_out/r5_gputests5.txt
timeout -k 10 300 python tools/small_j_probe.py --iters 60 > gpurun_out/r5_small_probe5.txt 2>&1 || { tail gpurun_out/r5_small_probe5.txt; exit 2; }
grep "update call" gpurun_out/r5_small_probe5.txt
CESX_NO_SMALL_PATH=1 timeout -k 10 300 python tools/small_j_probe.py --iters 60 > gpurun_out/r5_small_probe5_off.txt 2>&1
grep "update call" gpurun_out/r5_small_probe5_off.txt
timeout -k 10 300 python - > gpurun_out/r5_c4_5.txt 2>&1 <<'PY' || { tail -20 gpurun_out/r5_c4_5.txt; exit 6; }
import json, sys, os
sys.path.insert(0, '.')
import bench
from ces_amd import engine
for env in ("", "1"):
    if env: os.environ["CESX_NO_SMALL_PATH"] = "1"
    r = bench.engine_leg(engine, "C4 update only", 64, 50, 8192, "float32", 40, 0, prewarm_s=0.4)
    print("NO_SMALL_PATH=%r" % env, round(r["ms_per_step"], 4), {k: v["avg_launch_ms"] for k, v in r["roofline"]["kernels"].items()})
    r = bench.engine_leg(engine, "p256 J768", 256, 50, 768, "float32", 40, 0, prewarm_s=0.4)
    print("   (256,50,768)", round(r["ms_per_step"], 4))
    r = bench.engine_leg(engine, "p64 J512", 64, 50, 512, "float32", 40, 0, prewarm_s=0.4)
    print("   (64,50,512)", round(r["ms_per_step"], 4))
PY
cat gpurun_out/r5_c4_5.txt
timeout -k 10 400 bash tools/ab_lib.sh ces_amd/libcesx_r4.so 2 > gpurun_out/r5_ab_lib5.txt 2>&1 || exit 4
cat gpurun_out/r5_ab_lib5.txt
