#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/small_j_probe.py --iters 60 > gpurun_out/r5_small_probe.txt 2>&1 || { tail gpurun_out/r5_small_probe.txt; exit 2; }
cat gpurun_out/r5_small_probe.txt
for v in "eks none" "aldi constant" "aldi none dense"; do
  tag=$(echo $v | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/vt_$tag -- python3 tools/variant_trace.py $v > gpurun_out/vt_$tag.log 2>&1 || { tail gpurun_out/vt_$tag.log; exit 3; }
  python3 tools/trace_step.py gpurun_out/vt_$tag > gpurun_out/vt_$tag.txt 2>&1 || true
done
timeout -k 10 300 python - > gpurun_out/r5_variants4.txt 2>&1 <<'PY' || { tail -20 gpurun_out/r5_variants4.txt; exit 6; }
import json, sys
sys.path.insert(0, '.')
import bench
from ces_amd import engine
bench.VARIANTS = tuple(v for v in bench.VARIANTS if v[0] in ("aldi_default", "dense_sigma", "dense_gamma_sigma", "dense_gamma"))
d = bench.variants_leg(engine, 256, 256, 65536, "float32", 0)
print({k: (v.get('ms_per_step'), v.get('ratio_to_default'), v.get('error')) for k, v in d.items() if isinstance(v, dict)})
PY
cat gpurun_out/r5_variants4.txt
