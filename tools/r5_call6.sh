#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export CESX_BENCH_PREWARM_S=0.3
ARGS="bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr6_new -- python3 $ARGS > gpurun_out/tr6_new.log 2>&1 || exit 2
CESX_LIB=$PWD/ces_amd/libcesx_r4.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr6_r4 -- python3 $ARGS > gpurun_out/tr6_r4.log 2>&1 || exit 3
for t in new r4; do
  python3 tools/trace_step.py gpurun_out/tr6_$t 2 > gpurun_out/tr6_$t.txt 2>&1
  f=$(ls gpurun_out/tr6_$t/*/*_kernel_stats.csv | tail -1); cp $f gpurun_out/tr6_${t}_stats.csv
done
unset CESX_BENCH_PREWARM_S
timeout -k 10 400 bash tools/ab_lib.sh ces_amd/libcesx_r4.so 2 > gpurun_out/r5_ab_lib6.txt 2>&1 || exit 4
cat gpurun_out/r5_ab_lib6.txt
