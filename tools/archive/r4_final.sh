#!/bin/bash
# the round's final bench lines (after the counter profiles were committed under profiles/): tools/r4_final.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== bench (driver settings)"; python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "== bench (defaults)";        python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "== kernel stats";  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-config-legs > $O/bench_under_rocprof.json 2> $O/stats.err
cd $R
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/stats_kernel_stats.csv \;
rm -rf $O/stats
for W in c1_1e5_k3 c2_1e7_k5 c4_1e8_k10 c5_2.5e8_depth_k5; do echo "== $W"; python3 $R/bench.py --workload $W --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err || true; done
python3 - <<PY
import json
d = json.load(open("$O/bench_driver.json"))
print("driver line: value %.4g, %.5f ms/step, roofline frac %.3f (raw %.3f), eight chains %.4g, recorded %.4g, three chains %.4g" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_raw"] or 0, d["eight_chains_one_gpu_batched"]["value"], d["recorded"]["value"], d["three_chains_one_gpu"]["value"]))
PY
