cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out &&
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_update_kernels or medium_sizes or fault or recover" > gpurun_out/r5_t_small.txt 2>&1; tail -3 gpurun_out/r5_t_small.txt;
python3 - <<'PY' > gpurun_out/c4_f64.txt 2>&1
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ces_amd import engine
for rnd in range(2):
    for dt in ("float32", "float64"):
        r = bench.engine_leg(engine, "C4 update only", 64, 50, 8192, dt, 200, 0, prewarm_s=0.4)
        k = r["roofline"]["kernels"]
        print("%s: C4 %.4f ms/step  K1 %.4f K3 %.4f" % (dt, r["ms_per_step"], k["gram_kernel(K1)"]["avg_launch_ms"], k["update_kernel(K3)"]["avg_launch_ms"]), flush=True)
PY
cat gpurun_out/c4_f64.txt;
rocprofv3 --kernel-trace -d gpurun_out/prof_c4 -o c4 --output-format csv -- python3 bench.py --J 8192 --p 64 --n 50 --no-cpu-baseline --no-extras --steps 50 --warmup 5 > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err &&
python3 tools/trace_step.py gpurun_out/prof_c4 1 > gpurun_out/c4_trace.txt && rm -rf gpurun_out/prof_c4 && cat gpurun_out/c4_trace.txt
