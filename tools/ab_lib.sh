#!/bin/bash
# dev: A/B of two builds of libcesx.so (separate processes, alternating):  bash tools/ab_lib.sh LIB_B [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
B=$1; R=${2:-3}
for r in $(seq 1 $R); do
  for lib in ces_amd/libcesx.so $B; do
    CESX_LIB=$PWD/$lib CESX_BENCH_PREWARM_S=1.0 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', 'ms/step %.4f median %.4f K1 %.4f K3 %.4f gap %.4f' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernels']['gram_kernel(K1)']['avg_launch_ms'], d['roofline']['kernels']['update_kernel(K3)']['avg_launch_ms'], d['sampled_step']['gram_end_to_k3_start_ms'] or 0))"
  done
done
