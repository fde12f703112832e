set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sweeps_match_checker or many_states or many_wrong" > gpurun_out/r5_wl_t1.log 2>&1 || { tail -30 gpurun_out/r5_wl_t1.log; exit 1; }
tail -2 gpurun_out/r5_wl_t1.log
timeout -k 10 600 python tools/fuzz_parity.py 200 18 wide > gpurun_out/r5_wl_fuzz1.txt 2>&1 || { tail -15 gpurun_out/r5_wl_fuzz1.txt; exit 1; }
tail -1 gpurun_out/r5_wl_fuzz1.txt
export HML_TIME_NO_COMPAT=1
timeout -k 10 900 python tools/time_wide.py 20 40 64 > gpurun_out/r5_wl_time10.txt 2>&1 || { tail -15 gpurun_out/r5_wl_time10.txt; exit 1; }
cat gpurun_out/r5_wl_time10.txt
