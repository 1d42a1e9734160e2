#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5_gputests12.txt 2>&1 || { grep -E "^FAILED|^ERROR" gpurun_out/r5_gputests12.txt; }
tail -2 gpurun_out/r5_gputests12.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python bench.py > gpurun_out/r5_bench12.json 2> gpurun_out/r5_bench12.err || { tail -20 gpurun_out/r5_bench12.err; exit 6; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench12.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_median'])
for k,v in d['roofline']['kernels'].items(): print(k, v['avg_launch_ms'], v['frac'], v['frac_executed'])
print({k:(v.get('ms_per_step'), v.get('ratio_to_default')) for k,v in d['extra']['variants'].items() if isinstance(v,dict)})
for c in ('C5','C2_f64','C4'):
    print(c, d['extra'][c]['ms_per_step'], {k:(v['avg_launch_ms'],v['frac']) for k,v in d['extra'][c]['roofline']['kernels'].items()})
print('small', d['extra']['small_J']['cases'])
print('sharded', {k:(v.get('ms_per_step'), v.get('rccl_nranks')) for k,v in d['extra']['sharded_one_rank'].items() if isinstance(v,dict)})
print('e2e', {k:(v['ms_per_step']) for k,v in d['e2e'].items()})
print(d['e2e']['host_arrays']['update_call_ms_pct'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
