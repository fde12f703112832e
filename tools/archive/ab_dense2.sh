#!/bin/bash
# tools/ab_dense2.sh "<lib tag> [ENV=VAL ...]" ... : dense-regime kernel times per configuration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  set -- $cfg
  v=$1; shift
  i=$((i+1))
  rm -rf $R/gpurun_out/abd_$i
  echo "== $v $@"
  env HML_LIBRARY=$R/hammlet_amd/libhammlet_hip_$v.so "$@" timeout -k 10 120 python3 $R/tools/time_dense.py c3u 12 2>&1 | grep "ms/sweep"
done
