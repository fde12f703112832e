#!/bin/bash
# A/B of two library builds on the config-3 trace: rocprofv3 kernel durations + the script's own event brackets
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  export HML_LIBRARY=$R/hammlet_amd/libhammlet_hip_$v.so
  rm -rf $R/gpurun_out/ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -o run -- python3 $R/tools/time_fused_T.py 5 1e8 > $R/gpurun_out/ab_$v.log 2>&1
  tail -2 $R/gpurun_out/ab_$v.log
  python3 $R/tools/kstats.py $R/gpurun_out/ab_$v | head -12
done
