#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5_gputests13.txt 2>&1 || { grep -E "^FAILED|^ERROR" gpurun_out/r5_gputests13.txt; }
tail -2 gpurun_out/r5_gputests13.txt
bash tools/run_profiles.sh r05 C2 && bash tools/run_profiles.sh r05_c5 C5
# the line with its own traffic table (run_profiles.sh's bench ran before the table was regenerated)
cp gpurun_out/prof_out/traffic.json profiles/traffic.json
/usr/bin/time -v python bench.py > gpurun_out/prof_out/r05_bench.json 2> gpurun_out/r05_bench_final.err
grep -E "Elapsed|Maximum resident" gpurun_out/r05_bench_final.err
python bench.py --config C5 --no-extras > gpurun_out/prof_out/r05_c5_bench.json 2>> gpurun_out/r05_bench_final.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/prof_out/r05_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["frac_executed"], r["mfma_util"], r["traffic"], d["clock"]["k3_ghz"])
print({k:(v["avg_launch_ms"], v["frac"], v["frac_executed"], v["mfma_util"], v["traffic"]) for k,v in r["kernels"].items()})
PY
