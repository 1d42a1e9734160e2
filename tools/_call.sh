cd $GRAFT_REPO_ROOT/tools && mkdir -p ../gpurun_out && (
for b in base nodma noshift nobar nodma_noshift clk; do echo "== $b"; timeout -k 5 120 ./gram2_bench_$b 2 224 f64 | grep -v "^    type\|differ" ; done;
echo "== f64 first launch"; for b in base nodma clk; do echo "== $b"; timeout -k 5 120 ./gram2_bench_$b 1 256 f64 | grep -v "differ"; done ) > ../gpurun_out/r5_gram64_abl.txt 2>&1; cat ../gpurun_out/r5_gram64_abl.txt
