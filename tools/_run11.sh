cd $GRAFT_REPO_ROOT
python tools/_probe_k20.py 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -x -q -k "batched or attached or capacity" > gpurun_out/r5_t11.log 2>&1; tail -3 gpurun_out/r5_t11.log
python -m pytest tests/test_gpu_fuzz.py -x -q -k batched > gpurun_out/r5_t11b.log 2>&1; tail -3 gpurun_out/r5_t11b.log
for n in 8 16; do echo "chains $n"; python tools/multi_chain.py $n 1000 c3_1e8_k5_dynamic attached 2>&1 | tail -1; done
O=$GRAFT_REPO_ROOT/gpurun_out/chains16_prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
REPS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o att -- python3 $GRAFT_REPO_ROOT/tools/multi_chain.py 16 600 c3_1e8_k5_dynamic attached > $O/prof.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
head -9 $O/kernel_stats.txt
