#!/bin/bash
# The round's measurement pass on the GPU box (one gpurun call):  tools/round_profiles.sh <tag>
#   1. bench.py as the driver runs it (--steps 20 --warmup 5) and with its defaults
#   2. rocprofv3 --kernel-trace --stats of the default bench command
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the compressed legs, and of the dense regime (tools/time_dense.py c3u)
# Everything lands under gpurun_out/<tag>/; tools/pmc_summary.py + tools/kstats.py turn it into the files kept in profiles/.
set -e
TAG=${1:-r4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
if [ "${2:-all}" != "2" ]; then rm -rf $O; fi
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q='simple_timer\|generateRocpd\|tool.cpp\|^$'

PART=${2:-all}
if [ "$PART" != "2" ]; then
echo "== bench (driver settings)"; python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "== bench (defaults)";        python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "== kernel stats";  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-config-legs > $O/bench_under_rocprof.json 2> $O/stats.err
echo "== pmc fetch";     HML_BENCH_NO_TORCH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-uncompressed-leg --no-config-legs > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "== pmc write";     HML_BENCH_NO_TORCH=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-uncompressed-leg --no-config-legs > $O/pmc_write.json 2> $O/pmc_write.err
echo "== c4 pmc fetch";  HML_BENCH_NO_TORCH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_fetch -o run -- python3 $R/bench.py --workload c4_1e8_k10 --steps 200 --warmup 64 --no-cpu-baseline --no-stream-leg --no-two-chain-leg > $O/c4_fetch.json 2> $O/c4_fetch.err
echo "== c4 pmc write";  HML_BENCH_NO_TORCH=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4_write -o run -- python3 $R/bench.py --workload c4_1e8_k10 --steps 200 --warmup 64 --no-cpu-baseline --no-stream-leg --no-two-chain-leg > $O/c4_write.json 2> $O/c4_write.err
fi
if [ "$PART" = "1" ]; then
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-uncompressed-leg --no-config-legs" > /dev/null
python3 tools/pmc_summary.py $O/c4_fetch $O/c4_write $O/pmc_hbm_traffic_c4.json "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --workload c4_1e8_k10 --steps 200 --warmup 64 --no-cpu-baseline --no-stream-leg --no-two-chain-leg" > /dev/null
python3 tools/kstats.py $O/stats > $O/kernel_stats_c3_bench.txt
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/stats_kernel_stats.csv \;
rm -rf $O/stats
find $O/pmc_fetch $O/pmc_write $O/c4_fetch $O/c4_write -name '*.csv' -size +8M -delete
cat $O/bench_driver.json; head -14 $O/kernel_stats_c3_bench.txt 2>/dev/null || true
exit 0
fi
echo "== dense stats";   rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense_stats -o run -- python3 $R/tools/time_dense.py c3u 20 > $O/dense_c3u.txt 2> $O/dense_stats.err
echo "== dense pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dense_fetch -o run -- python3 $R/tools/time_dense.py c3u 20 > /dev/null 2> $O/dense_fetch.err
echo "== dense pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dense_write -o run -- python3 $R/tools/time_dense.py c3u 20 > /dev/null 2> $O/dense_write.err
echo "== c5 pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5_fetch -o run -- python3 $R/tools/time_dense.py c5 10 > /dev/null 2> $O/c5_fetch.err
echo "== c5 pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5_write -o run -- python3 $R/tools/time_dense.py c5 10 > /dev/null 2> $O/c5_write.err
echo "== c5 stats";      rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o run -- python3 $R/tools/time_dense.py c5 20 > $O/dense_c5.txt 2> $O/c5_stats.err

cd $R
if [ "$PART" = "all" ]; then python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-uncompressed-leg" > /dev/null; fi
python3 tools/pmc_summary.py $O/dense_fetch $O/dense_write $O/pmc_hbm_traffic_c3u.json "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 tools/time_dense.py c3u 20" > /dev/null
python3 tools/pmc_summary.py $O/c5_fetch $O/c5_write $O/pmc_hbm_traffic_c5.json "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 tools/time_dense.py c5 10" > /dev/null
if [ "$PART" = "all" ]; then python3 tools/kstats.py $O/stats > $O/kernel_stats_c3_bench.txt; fi
python3 tools/kstats.py $O/dense_stats > $O/kernel_stats_c3u.txt
python3 tools/kstats.py $O/c5_stats > $O/kernel_stats_c5.txt
for d in stats dense_stats c5_stats; do [ -d $O/$d ] && find $O/$d -name '*kernel_stats.csv' -exec cp {} $O/${d}_kernel_stats.csv \; ; done
# the raw traces are large: keep the summaries only
rm -rf $O/stats $O/dense_stats $O/c5_stats
find $O -name '*.csv' -size +8M -delete
# SQ issue / wait counters of the dense regime's first pass (two passes of their own)
cd /tmp
for W in c3u c5; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq_a_$W -o run -- python3 $R/tools/time_dense.py $W 6 > /dev/null 2> $O/sq_a_$W.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/sq_b_$W -o run -- python3 $R/tools/time_dense.py $W 6 > /dev/null 2> $O/sq_b_$W.err
python3 - > $O/sq_counters_$W.txt <<PY
import csv, glob, collections
print("# rocprofv3 --pmc, two passes (tools/round_profiles.sh): medians per launch over the settled sweeps of tools/time_dense.py $W")
for d in ("sq_a_$W", "sq_b_$W"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        if any(t in k for t in ("trellis_rows", "counts_dense", "trellis_states", "compact_scan", "compact_scatter", "trellis_refit")):
            print(k, {c: "%.4g" % sorted(v)[len(v) // 2] for c, v in acc[k].items()})
PY
rm -rf $O/sq_a_$W $O/sq_b_$W
done
cd $R
# the other BASELINE configurations through the same script, end-to-end command lines, chains per GPU
for W in c1_1e5_k3 c2_1e7_k5 c4_1e8_k10 c5_2.5e8_depth_k5; do python3 $R/bench.py --workload $W --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err || true; done
bash $R/tools/e2e_cli.sh > $O/cli_end_to_end_1e8.log 2>&1 || true
bash $R/tools/e2e_small.sh > $O/cli_end_to_end_small.log 2>&1 || true
cat $O/bench_driver.json; cat $O/dense_c3u.txt $O/dense_c5.txt; head -14 $O/kernel_stats_c3_bench.txt 2>/dev/null || true
