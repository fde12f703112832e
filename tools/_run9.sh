set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "batched or attached or capacity" > gpurun_out/r5_t9.log 2>&1; tail -3 gpurun_out/r5_t9.log
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_compat.py -x -q > gpurun_out/r5_t9b.log 2>&1; tail -3 gpurun_out/r5_t9b.log
for n in 8 16; do
  for sp in 1 0; do echo "chains $n split $sp"; HML_FM_SPLIT=$sp python tools/multi_chain.py $n 1000 c3_1e8_k5_dynamic attached 2>&1 | tail -2; done
done
BURNIN=100 python tools/compat_time.py c3_1e8_k5_dynamic 24 2>&1 | tail -1
