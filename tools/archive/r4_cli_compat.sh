#!/bin/bash
# End to end: `hammlet -compat` (the reference's own chain, bit for bit) and the default mode on config 3's trace (raw float32 input),
# 200 sweeps, every 10th recorded, marginals written.   tools/r4_cli_compat.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
RAW=/tmp/hml_c3.f32
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
import bench, hammlet_amd
T, K, levels, sigma, dwell, seed = bench.WORKLOADS["c3_1e8_k5_dynamic"]
hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, seed, nthreads=8).tofile("$RAW")
PY
for mode in "-compat" ""; do
  for rep in 1 2; do
    start=$(date +%s.%N)
    $ROOT/hammlet_amd/hammlet $mode -raw $RAW -a -s 5 -R 1 -i F 200 10 -w -o /tmp/hml_c3_out- .csv -O marginals > /dev/null || exit 1
    end=$(date +%s.%N)
    python3 -c "print('hammlet %-8s -i F 200 10 on 10^8 positions: %.2f s wall clock' % ('$mode' or '(default)', $end - $start))"
  done
  wc -l /tmp/hml_c3_out-marginals.csv
done
rm -f $RAW /tmp/hml_c3_out-*
