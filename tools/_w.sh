set -e
export HML_TIME_NO_COMPAT=1
for nw in 2 4; do
  export HML_WIDE_NW=$nw
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sweeps_match_checker or many_states" > gpurun_out/r5_wl_nw$nw.log 2>&1 || { tail -30 gpurun_out/r5_wl_nw$nw.log; exit 1; }
  tail -1 gpurun_out/r5_wl_nw$nw.log
  timeout -k 10 600 python tools/fuzz_parity.py 100 2$nw wide > gpurun_out/r5_wl_fuzz_nw$nw.txt 2>&1 || { tail -15 gpurun_out/r5_wl_fuzz_nw$nw.txt; exit 1; }
  tail -1 gpurun_out/r5_wl_fuzz_nw$nw.txt
  echo "NW=$nw" >> gpurun_out/r5_wl_time8.txt
  timeout -k 10 900 python tools/time_wide.py 20 40 64 >> gpurun_out/r5_wl_time8.txt 2>&1 || { tail -15 gpurun_out/r5_wl_time8.txt; exit 1; }
done
cat gpurun_out/r5_wl_time8.txt
