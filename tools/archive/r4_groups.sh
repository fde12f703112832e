#!/bin/bash
# chain groups inside hml_iterate_many (HML_MANY_GROUPS) on one box:  tools/r4_groups.sh [<dev lib tag>]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
[ -n "$1" ] && export HML_LIBRARY=$ROOT/hammlet_amd/libhammlet_hip_k5$1.so
for cfg in "2 1" "2 2" "4 1" "4 2" "8 1" "8 2" "8 4" "16 1" "16 2" "16 4" "32 2"; do
  set -- $cfg
  echo "== chains $1 groups $2"
  HML_MANY_GROUPS=$2 REPS=3 timeout -k 10 300 python tools/multi_chain.py $1 1000 c3_1e8_k5_dynamic attached || exit 1
done
