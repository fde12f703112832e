set -e
export HML_TIME_NO_COMPAT=1
timeout -k 10 600 python tools/time_wide.py 20 > gpurun_out/r5_wl_time2.txt 2>&1 || { tail -15 gpurun_out/r5_wl_time2.txt; exit 1; }
HML_COMPAT_WARMUP=32 timeout -k 10 600 python tools/time_wide.py 20 64 >> gpurun_out/r5_wl_time2.txt 2>&1
HML_COMPAT_WARMUP=32 HML_WIDE_L=16 timeout -k 10 600 python tools/time_wide.py 20 >> gpurun_out/r5_wl_time2.txt 2>&1
HML_COMPAT_WARMUP=32 HML_WIDE_L=64 timeout -k 10 600 python tools/time_wide.py 20 >> gpurun_out/r5_wl_time2.txt 2>&1
cat gpurun_out/r5_wl_time2.txt
