#!/bin/bash
# tools/refit_ab.sh <out dir>: durations per repair round (4 rounds), then the sweep with 4 / 3 / 2 / 1 parallel rounds, C3u and C5
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1
mkdir -p $O
export HML_LIBRARY=$R/hammlet_amd/libhammlet_hip_k5rr.so
$R/tools/refit_probe.sh $1 c3u 60
$R/tools/refit_probe.sh $1 c5 40
cd /tmp
for W in c3u c5; do for r in 4 3 2 1 4 2; do echo "== $W rounds $r" | tee -a $O/rounds.txt; HML_TRELLIS_REFIT_ROUNDS=$r python3 $R/tools/time_dense.py $W 40 2>&1 | tee -a $O/rounds.txt; done; done
