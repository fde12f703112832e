#!/bin/bash
# rocprofv3 kernel stats of the dense regime: tools/prof_dense.sh c3u|c5 [sweeps]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/dense_$1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dense_$1 -o run -- python3 $R/tools/time_dense.py $1 ${2:-20} 2>&1 | grep -v "simple_timer\|generateRocpd\|tool.cpp"
python3 $R/tools/kstats.py $R/gpurun_out/dense_$1 2>/dev/null | head -${3:-16}
