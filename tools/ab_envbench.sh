#!/bin/bash
# dev: A/B of environment switches read at cesx_create, separate bench processes alternating:  bash tools/ab_envbench.sh "VAR=a" "VAR=b" [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for cfg in "$A" "$B"; do
    env $cfg CESX_BENCH_PREWARM_S=1.0 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', 'ms/step %.4f median %.4f K1 %.4f K3 %.4f gap %.4f' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernels']['gram_kernel(K1)']['avg_launch_ms'], d['roofline']['kernels']['update_kernel(K3)']['avg_launch_ms'], d['sampled_step']['gram_end_to_k3_start_ms'] or 0))"
  done
done
