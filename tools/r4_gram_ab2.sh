#!/bin/bash
# dev: A/B of Gram kernel build variants on the GPU box:  bash tools/r4_gram_ab2.sh "<variant suffixes>" [f64]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r4_gram_ab2.txt
: > $O
for rep in 1 2; do
for v in $1; do
  [ "$v" = "-" ] && v=""
  echo "== variant '$v'" >> $O
  timeout -k 10 120 tools/gram2_bench$v 1 256 $2 2>&1 | grep "LDS-DMA" >> $O || exit 2
  timeout -k 10 120 tools/gram2_bench$v 2 ${3:-248} $2 2>&1 | grep "LDS-DMA" >> $O || exit 2
done
done
cat $O
