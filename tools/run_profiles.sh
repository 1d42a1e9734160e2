#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root):  bash tools/run_profiles.sh TAG CONFIG
#   kernel trace + stats, then one rocprofv3 --pmc pass per counter (the guide's rule: counters in their own runs).
# Results land under gpurun_out/prof_${TAG}_*; condense them here with tools/prof_summary.py.
set -o pipefail
TAG=$1; CFG=${2:-C2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
ARGS="bench.py --config $CFG --no-cpu-baseline --no-extras --steps 12 --warmup 3"
export CESX_BENCH_PREWARM_S=${CESX_BENCH_PREWARM_S:-0.3}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_k -- python3 $ARGS > gpurun_out/prof_${TAG}_k.log 2>&1 || exit 2
# (--pmc serialises the streams' kernels: a kernel of the caller's stream that polls for one of the side stream would wait
#  for its time-out -- the engine recovers, bench.py carries on, but the event join is the honest mode for these passes)
export CESX_POLL_JOIN=0
for ctr in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/prof_${TAG}_$ctr -- python3 $ARGS > gpurun_out/prof_${TAG}_$ctr.log 2>&1 || exit 3
done
unset CESX_POLL_JOIN
# condense on the box (gpurun merges at most 64 MiB back, the raw counter CSVs are larger): the summaries land in profiles/ of the
# box's copy of the tree and are copied to gpurun_out/prof_out/; the raw directories are removed
python3 tools/prof_summary.py ${TAG}:${CFG} gpurun_out/prof_${TAG}_k gpurun_out/prof_${TAG}_FETCH_SIZE gpurun_out/prof_${TAG}_WRITE_SIZE \
    gpurun_out/prof_${TAG}_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/prof_${TAG}_GRBM_GUI_ACTIVE > gpurun_out/prof_${TAG}_summary.log 2>&1 || exit 4
# the benchmark line LAST: profiles/traffic.json now carries this tree's kernel-source hash, so the line takes its `traffic` from it
if [ "$CFG" = "C2" ]; then python3 bench.py --config $CFG > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_${TAG}_bench.err
else python3 bench.py --config $CFG --no-extras > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_${TAG}_bench.err; fi
mkdir -p gpurun_out/prof_out
cp profiles/${TAG}_* profiles/traffic.json gpurun_out/prof_out/
cp gpurun_out/prof_${TAG}_bench.json gpurun_out/prof_out/${TAG}_bench.json
rm -rf gpurun_out/prof_${TAG}_k gpurun_out/prof_${TAG}_FETCH_SIZE gpurun_out/prof_${TAG}_WRITE_SIZE gpurun_out/prof_${TAG}_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/prof_${TAG}_GRBM_GUI_ACTIVE
echo done $TAG $CFG
