#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r5_gram_opq2.txt
: > $O
for v in _old _q1 _q3 _q8 _q9 _q11 _q31 _q1 _q8 _q9 _old; do
  echo "== variant '$v'" >> $O
  timeout -k 10 120 tools/gram2_bench$v 1 256 2>&1 | grep -A1 "^f32" | grep staged >> $O || exit 2
  timeout -k 10 120 tools/gram2_bench$v 2 248 2>&1 | grep -A1 "^f32" | grep staged >> $O || exit 2
  timeout -k 10 200 tools/gram2_bench$v 1 256 f64 2>&1 | grep -A1 "^f64" | grep staged >> $O || exit 3
  timeout -k 10 200 tools/gram2_bench$v 2 224 f64 2>&1 | grep -A1 "^f64" | grep staged >> $O || exit 3
done
echo "gram variants done"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests3.txt 2>&1 || { tail -40 gpurun_out/r5_gputests3.txt; exit 5; }
tail -3 gpurun_out/r5_gputests3.txt
timeout -k 10 300 python - > gpurun_out/r5_variants3.txt 2>&1 <<'PY' || { tail -20 gpurun_out/r5_variants3.txt; exit 6; }
import json, sys
sys.path.insert(0, '.')
import bench
from ces_amd import engine
print(json.dumps(bench.variants_leg(engine, 256, 256, 65536, "float32", 0), indent=0))
PY
grep -E "ms_per_step|ratio|^\"[a-z_]+\":" gpurun_out/r5_variants3.txt | paste - - - | cut -c1-150
