cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_full_golden.py tests/test_gpu_reference_bridge.py -x -q > gpurun_out/r5_t10.log 2>&1; tail -5 gpurun_out/r5_t10.log
O=$GRAFT_REPO_ROOT/gpurun_out/chains16_prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
REPS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o att -- python3 $GRAFT_REPO_ROOT/tools/multi_chain.py 16 600 c3_1e8_k5_dynamic attached > $O/prof.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
head -14 $O/kernel_stats.txt
BURNIN=100 bash $GRAFT_REPO_ROOT/tools/compat_prof.sh c3_1e8_k5_dynamic 24 2>&1 | tail -6
