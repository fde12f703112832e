set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sweeps_match_checker or many_states or many_wrong" > gpurun_out/r5_wl_t1.log 2>&1 || { tail -30 gpurun_out/r5_wl_t1.log; exit 1; }
tail -3 gpurun_out/r5_wl_t1.log
timeout -k 10 600 python tools/fuzz_parity.py 200 17 wide > gpurun_out/r5_wl_fuzz1.txt 2>&1 || { tail -15 gpurun_out/r5_wl_fuzz1.txt; exit 1; }
tail -1 gpurun_out/r5_wl_fuzz1.txt
export TMPDIR=/tmp HML_TIME_NO_COMPAT=1
O=$GRAFT_REPO_ROOT/gpurun_out/wl_stats2
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o run -- python3 tools/time_wide.py 20 40 64 > $O/k.out 2> $O/k.err || { tail -5 $O/k.err; exit 1; }
python3 tools/kstats.py $O/k > $O/kernel_stats_wide.txt
cat $O/k.out >> $O/kernel_stats_wide.txt
find $O -name '*trace.csv' -size +4M -delete
head -12 $O/kernel_stats_wide.txt; tail -3 $O/kernel_stats_wide.txt
