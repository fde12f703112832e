#!/bin/bash
# rocprofv3 kernel summary of a reference-compatible run:  tools/compat_prof.sh [workload] [sweeps]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/compat_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o att -- python3 $ROOT/tools/compat_time.py ${1:-c2_1e7_k5} ${2:-5} > $OUT/prof.log 2>&1
python3 $ROOT/tools/kstats.py $OUT/prof > $OUT/kernel_stats.txt 2>&1
rm -rf $OUT/prof
head -12 $OUT/kernel_stats.txt
