set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_final
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver settings done"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "defaults done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --no-cpu-baseline --no-config-legs > $O/bench_under_rocprof.json 2> $O/stats.err; echo "rocprof done"
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/stats_kernel_stats.csv \;
python3 tools/kstats.py $O/stats > $O/kernel_stats_c3_bench.txt
rm -rf $O/stats
head -12 $O/kernel_stats_c3_bench.txt
