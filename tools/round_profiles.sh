#!/bin/bash
# The round's measurement pass on the GPU box (one gpurun call):  tools/round_profiles.sh <tag>
#   1. bench.py as the driver runs it (--steps 20 --warmup 5) and with its defaults
#   2. rocprofv3 --kernel-trace --stats of the default bench command
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the compressed legs, and of the dense regime (tools/time_dense.py c3u)
# Everything lands under gpurun_out/<tag>/; tools/pmc_summary.py + tools/kstats.py turn it into the files kept in profiles/.
set -e
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q='simple_timer\|generateRocpd\|tool.cpp\|^$'

echo "== bench (driver settings)"; python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "== bench (defaults)";        python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "== kernel stats";  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
echo "== pmc fetch";     rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-uncompressed-leg > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "== pmc write";     rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-uncompressed-leg > $O/pmc_write.json 2> $O/pmc_write.err
echo "== dense stats";   rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense_stats -o run -- python3 $R/tools/time_dense.py c3u 20 > $O/dense_c3u.txt 2> $O/dense_stats.err
echo "== dense pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dense_fetch -o run -- python3 $R/tools/time_dense.py c3u 20 > /dev/null 2> $O/dense_fetch.err
echo "== dense pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dense_write -o run -- python3 $R/tools/time_dense.py c3u 20 > /dev/null 2> $O/dense_write.err
echo "== c5 pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5_fetch -o run -- python3 $R/tools/time_dense.py c5 10 > /dev/null 2> $O/c5_fetch.err
echo "== c5 pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5_write -o run -- python3 $R/tools/time_dense.py c5 10 > /dev/null 2> $O/c5_write.err
echo "== c5 stats";      rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o run -- python3 $R/tools/time_dense.py c5 20 > $O/dense_c5.txt 2> $O/c5_stats.err

cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-uncompressed-leg" > /dev/null
python3 tools/pmc_summary.py $O/dense_fetch $O/dense_write $O/pmc_hbm_traffic_c3u.json "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 tools/time_dense.py c3u 20" > /dev/null
python3 tools/pmc_summary.py $O/c5_fetch $O/c5_write $O/pmc_hbm_traffic_c5.json "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 tools/time_dense.py c5 10" > /dev/null
python3 tools/kstats.py $O/stats > $O/kernel_stats_c3_bench.txt
python3 tools/kstats.py $O/dense_stats > $O/kernel_stats_c3u.txt
python3 tools/kstats.py $O/c5_stats > $O/kernel_stats_c5.txt
for d in stats dense_stats c5_stats; do find $O/$d -name '*kernel_stats.csv' -exec cp {} $O/${d}_kernel_stats.csv \; ; done
# the raw traces are large: keep the summaries only
rm -rf $O/stats $O/dense_stats $O/c5_stats
find $O/pmc_fetch $O/pmc_write $O/dense_fetch $O/dense_write $O/c5_fetch $O/c5_write -name '*.csv' -size +8M -delete
cat $O/bench_driver.json; cat $O/dense_c3u.txt $O/dense_c5.txt; head -14 $O/kernel_stats_c3_bench.txt
