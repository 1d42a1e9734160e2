cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out &&
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "warm_start" > gpurun_out/r5_t_ns2.txt 2>&1; tail -25 gpurun_out/r5_t_ns2.txt
