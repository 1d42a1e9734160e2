cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out &&
python tools/variants_only.py eks time_step_constant > gpurun_out/r5_variants14.txt 2>&1; cat gpurun_out/r5_variants14.txt
