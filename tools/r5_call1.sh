#!/bin/bash
# dev (round 5): Gram kernel A/B (tree vs round-4 source), whole-step A/B of the two libraries, GPU tests
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r5_gram_ab.txt
: > $O
for v in "" _old; do
  echo "== variant '$v'" >> $O
  timeout -k 10 120 tools/gram2_bench$v 1 256 >> $O 2>&1 || exit 2
  timeout -k 10 120 tools/gram2_bench$v 2 248 >> $O 2>&1 || exit 2
  timeout -k 10 200 tools/gram2_bench$v 1 256 f64 >> $O 2>&1 || exit 3
  timeout -k 10 200 tools/gram2_bench$v 2 224 f64 >> $O 2>&1 || exit 3
done
echo "gram ab done"
timeout -k 10 400 bash tools/ab_lib.sh ces_amd/libcesx_r4.so 3 > gpurun_out/r5_ab_lib.txt 2>&1 || exit 4
echo "lib ab done"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests1.txt 2>&1 || { tail -30 gpurun_out/r5_gputests1.txt; exit 5; }
tail -3 gpurun_out/r5_gputests1.txt
