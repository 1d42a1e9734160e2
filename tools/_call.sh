cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out &&
python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err; tail -c 600 gpurun_out/r05_bench_final.json;
python bench.py --config C5 --no-extras > gpurun_out/r05_c5_bench_final.json 2> gpurun_out/r05_c5_bench_final.err; tail -c 300 gpurun_out/r05_c5_bench_final.json;
python tools/soak_long.py > gpurun_out/r05_soak2.txt 2>&1; tail -5 gpurun_out/r05_soak2.txt
