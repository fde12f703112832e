#!/bin/bash
# round 4: chains sharing one GPU - batched (private constructions) against attached (shared construction + many-chain block
# kernel), with the rocprofv3 kernel summary of the attached run.   usage: tools/r4_many.sh [chains] [workload]
set -e
R=${1:-8}
WL=${2:-c3_1e8_k5_dynamic}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4_many
mkdir -p $OUT
cd $ROOT
if [ -z "$SKIP_TIMING" ]; then python tools/multi_chain.py 1 1000 $WL threads > $OUT/chains.txt 2>&1
python tools/multi_chain.py $R 1000 $WL many >> $OUT/chains.txt 2>&1
python tools/multi_chain.py $R 1000 $WL attached >> $OUT/chains.txt 2>&1
cat $OUT/chains.txt; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o att -- python3 $ROOT/tools/multi_chain.py $R 300 $WL attached > $OUT/prof.log 2>&1
python3 $ROOT/tools/kstats.py $OUT/prof > $OUT/kernel_stats.txt 2>&1 || true
cat $OUT/kernel_stats.txt | head -30
