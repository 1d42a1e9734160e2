#!/bin/bash
# dev: A/B of the Gram kernels on the GPU box (run through gpurun from the repo root):  bash tools/r4_gram_ab.sh "<variant suffixes>"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r4_gram_ab.txt
: > $O
for v in $1; do
  [ "$v" = "-" ] && v=""
  echo "== variant '$v'" >> $O
  timeout -k 10 120 tools/gram2_bench$v 1 256 >> $O 2>&1 || exit 2
  timeout -k 10 120 tools/gram2_bench$v 2 248 >> $O 2>&1 || exit 2
done
echo "== f64 (C5 shape)" >> $O
timeout -k 10 200 tools/gram2_bench 1 256 f64 >> $O 2>&1 || exit 3
timeout -k 10 200 tools/gram2_bench 2 224 f64 >> $O 2>&1 || exit 3
