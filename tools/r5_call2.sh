#!/bin/bash
# dev (round 5): Gram kernel variants (which scalar tests are re-evaluated at their use), GPU tests, the full bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r5_gram_opq.txt
: > $O
for v in _old _q0 _q1 _q4 _q5 _q6 _q7 _old _q7 _q5; do
  echo "== variant '$v'" >> $O
  timeout -k 10 120 tools/gram2_bench$v 1 256 2>&1 | grep -A1 "^f32" | grep staged >> $O || exit 2
  timeout -k 10 120 tools/gram2_bench$v 2 248 2>&1 | grep -A1 "^f32" | grep staged >> $O || exit 2
  timeout -k 10 200 tools/gram2_bench$v 1 256 f64 2>&1 | grep -A1 "^f64" | grep staged >> $O || exit 3
  timeout -k 10 200 tools/gram2_bench$v 2 224 f64 2>&1 | grep -A1 "^f64" | grep staged >> $O || exit 3
done
echo "gram variants done"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests2.txt 2>&1 || { tail -30 gpurun_out/r5_gputests2.txt; exit 5; }
tail -3 gpurun_out/r5_gputests2.txt
timeout -k 10 900 python bench.py > gpurun_out/r5_bench2.json 2> gpurun_out/r5_bench2.err || { tail -20 gpurun_out/r5_bench2.err; exit 6; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernels'])
print(json.dumps(d['extra'].get('variants'), indent=0)[:3000])
print(json.dumps(d['extra'].get('sharded_one_rank'))[:1500])
PY
