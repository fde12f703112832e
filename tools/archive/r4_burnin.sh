#!/bin/bash
# warm-up policies of young chains:  tools/r4_burnin.sh
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "64 16" "0 8" "16 8" "0 4" "0 2"; do
  set -- $cfg
  echo "== burn-in floor for $1 sweeps, $2 quiet sweeps per step"
  HML_FWD_BURNIN_SWEEPS=$1 HML_FWD_QUIET=$2 python3 tools/r4_burnin.py
done
