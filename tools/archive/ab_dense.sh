#!/bin/bash
# per-kernel times of the dense regime for several library builds: tools/ab_dense.sh k5 k5skipP1 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  export HML_LIBRARY=$R/hammlet_amd/libhammlet_hip_$v.so
  rm -rf $R/gpurun_out/abd_$v
  echo "== $v"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abd_$v -o run -- python3 $R/tools/time_dense.py c3u 10 2>&1 | grep "ms/sweep"
  python3 $R/tools/kstats.py $R/gpurun_out/abd_$v 2>/dev/null | head -3
done
