#!/bin/bash
# tools/refit_probe.sh <out dir> <workload> [sweeps]: kernel trace of tools/time_dense.py, durations per repair round (tools/refit_rounds.py)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; W=$2; N=${3:-60}
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr_$W -o run -- python3 $R/tools/time_dense.py $W $N > $O/time_$W.txt 2> $O/tr_$W.err
python3 $R/tools/refit_rounds.py $O/tr_$W | tee $O/refit_rounds_$W.txt
rm -rf $O/tr_$W
