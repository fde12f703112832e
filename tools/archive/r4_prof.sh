#!/bin/bash
# rocprofv3 kernel summary of the 8-chain attached run with a development library:  tools/r4_prof.sh <lib tag> [chains] [sweeps]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
R=${2:-8}
N=${3:-300}
OUT=$ROOT/gpurun_out/r4_prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export HML_LIBRARY=$ROOT/hammlet_amd/libhammlet_hip_k5$TAG.so
[ -z "$TAG" -o "$TAG" = "full" ] && unset HML_LIBRARY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o att -- python3 $ROOT/tools/multi_chain.py $R $N c3_1e8_k5_dynamic attached > $OUT/prof.log 2>&1
python3 $ROOT/tools/kstats.py $OUT/prof > $OUT/kernel_stats.txt 2>&1
rm -rf $OUT/prof
head -12 $OUT/kernel_stats.txt
