#!/bin/bash
# `hammlet -chains N` on ONE GPU, config 3 (10^8 positions, 5 states, -i F 1000 10): wall clock of the whole command (round 4: the chains
# of a GPU share its construction and are driven in lockstep through hml_iterate_many)
cd ${GRAFT_REPO_ROOT:-.}
python3 - <<PY
import bench, hammlet_amd
T, K, levels, sigma, dwell, seed = bench.WORKLOADS["c3_1e8_k5_dynamic"]
hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, seed, nthreads=16).tofile("/tmp/c3.f32")
PY
for n in 1 2 4 8; do
  rm -f /tmp/o$n-*
  s=$(date +%s.%N)
  ./hammlet_amd/hammlet -raw /tmp/c3.f32 -a -s 5 -R 1 -i F 1000 10 -chains $n -o /tmp/o$n- .csv -O marginals parameters -w > /dev/null
  e=$(date +%s.%N)
  echo "hammlet -chains $n: $(echo "$e - $s" | bc 2>/dev/null || python3 -c "print($e - $s)") s wall clock, marginals $(wc -l < /tmp/o$n-marginals.csv) segments"
done
