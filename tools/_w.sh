cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_compat.py tests/test_gpu_fuzz.py tests/test_gpu_full_golden.py -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_reference_bridge.py -x -q 2>&1 | tail -2
python tools/time_wide.py 20 40 64 2>&1 | tail -3 | cut -c1-330
BURNIN=100 python tools/compat_time.py c3_1e8_k5_dynamic 24 2>&1 | tail -1
