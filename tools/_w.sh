set -e
export HML_TIME_NO_COMPAT=1
rm -f gpurun_out/r5_dense_thr.txt
for thr in 4194304 1048576 524288 262144; do echo "HML_DENSE_MIN_BLOCKS=$thr" >> gpurun_out/r5_dense_thr.txt; HML_DENSE_MIN_BLOCKS=$thr timeout -k 10 900 python tools/time_wide.py 5 8 10 12 16 >> gpurun_out/r5_dense_thr.txt 2>&1; done
cut -c1-300 gpurun_out/r5_dense_thr.txt
