#!/bin/bash
# dev: SQ counters of the Gram kernels (standalone bench tool) on the GPU box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4pmc
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/r4pmc/$tag -- tools/gram2_bench 2 248 > gpurun_out/r4pmc/$tag.log 2>&1 || exit 2
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r4pmc/*/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:40], r['Counter_Name'])
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, v in sorted(agg.items()):
        print(k[0], k[1], 'avg per dispatch %.4g' % (v[0] / v[1]), 'n', v[1])
PY
