R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python tools/time_dense.py c3u 20 2>&1 | tail -2
python tools/time_dense.py c5 10 2>&1 | tail -2
python -m pytest tests/test_gpu_fullsize_c4_c5.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "dense or fuzz_against or trellis or geometry" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
echo "== pmc fetch";     HML_BENCH_THREADS=8 HML_BENCH_NO_TORCH=1 timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-uncompressed-leg --no-config-legs > $O/pmc_fetch.json 2> $O/pmc_fetch.err; echo "rc $?"
