cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out &&
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_update_kernels" > gpurun_out/r5_t_small.txt 2>&1; tail -15 gpurun_out/r5_t_small.txt
