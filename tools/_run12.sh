cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_compat.py tests/test_gpu_full_golden.py -x -q > gpurun_out/r5_t12.log 2>&1; tail -3 gpurun_out/r5_t12.log
python -m pytest tests/test_gpu_fuzz.py -x -q -k compatible > gpurun_out/r5_t12b.log 2>&1; tail -3 gpurun_out/r5_t12b.log
BURNIN=100 python tools/compat_time.py c3_1e8_k5_dynamic 24 2>&1 | tail -1
BURNIN=100 bash tools/compat_prof.sh c3_1e8_k5_dynamic 24 2>&1 | tail -4
