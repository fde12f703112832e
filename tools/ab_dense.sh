#!/bin/bash
# A/B of development builds (tools/dev_build.py 5 <tag> -D...) on the dense regime:  tools/ab_dense.sh <out dir> <workload> <tag> [<tag> ...]
# per tag: tools/time_dense.py (ms per sweep, checksums of the chain) and the kernel statistics of the same command under rocprofv3
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; W=$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  export HML_LIBRARY=$R/hammlet_amd/libhammlet_hip_k5$t.so
  echo "== $t" | tee -a $O/ab.txt
  python3 $R/tools/time_dense.py $W 20 >> $O/ab.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$t -o run -- python3 $R/tools/time_dense.py $W 20 > /dev/null 2> $O/st_$t.err
  python3 $R/tools/kstats.py $O/st_$t | head -8 >> $O/ab.txt
  rm -rf $O/st_$t
done
cat $O/ab.txt
