#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5_gputests11.txt 2>&1 || { grep -E "^FAILED|^ERROR" gpurun_out/r5_gputests11.txt; }
tail -2 gpurun_out/r5_gputests11.txt
timeout -k 10 300 python - > gpurun_out/r5_variants11.txt 2>&1 <<'PY' || { tail -20 gpurun_out/r5_variants11.txt; exit 6; }
import json, sys
sys.path.insert(0, '.')
import bench
from ces_amd import engine
bench.VARIANTS = tuple(v for v in bench.VARIANTS if v[0] in ("aldi_default", "time_step_spectral"))
d = bench.variants_leg(engine, 256, 256, 65536, "float32", 0)
print({k: (v.get('ms_per_step'), v.get('ratio_to_default'), v.get('error')) for k, v in d.items() if isinstance(v, dict)})
PY
cat gpurun_out/r5_variants11.txt
