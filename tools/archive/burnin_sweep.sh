#!/bin/bash
# refits and sweep time of young chains under faster warm-up decay: tools/burnin_sweep.sh
for cfg in "256 32" "64 16" "32 8" "0 8" "0 4"; do
  set -- $cfg
  echo "== burn-in floor for $1 sweeps, $2 quiet sweeps per step"
  for wl in c3_1e8_k5_dynamic c2_1e7_k5; do
    echo $wl; HML_LIBRARY=$PWD/hammlet_amd/libhammlet_hip_k5.so HML_FWD_BURNIN_SWEEPS=$1 HML_FWD_QUIET=$2 PER=25 python3 tools/burnin_profile.py $wl 16
  done
  echo c4_1e8_k10; HML_LIBRARY=$PWD/hammlet_amd/libhammlet_hip_k10.so HML_FWD_BURNIN_SWEEPS=$1 HML_FWD_QUIET=$2 PER=25 python3 tools/burnin_profile.py c4_1e8_k10 16
done
