cd $GRAFT_REPO_ROOT
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_t13.log 2>&1; tail -5 gpurun_out/r5_t13.log
BURNIN=100 python tools/compat_time.py c3_1e8_k5_dynamic 24 2>&1 | tail -1
for g in 1 2 4; do echo "16 chains, groups $g"; HML_MANY_GROUPS=$g python tools/multi_chain.py 16 1000 c3_1e8_k5_dynamic attached 2>&1 | tail -1; done
