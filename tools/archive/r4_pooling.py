"""What the chain-parallel pooling costs at ONE rank and full size: export + ncclAllReduce + install
of the marginals of configs 3, 4 and 5 after 100 recorded sweeps - seconds and bytes, with the number of non-zero
difference entries the payload really carries.   python tools/r4_pooling.py [workloads...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, hammlet_amd
from hammlet_amd.capi import Pool
wls = sys.argv[1:] or ["c3_1e8_k5_dynamic", "c4_1e8_k10", "c5_2.5e8_depth_k5"]
pool = Pool(0, 0, 1, Pool.unique_id())
for wl in wls:
    T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS[wl]
    x = hammlet_amd.synth_depth(T, depth=dwell, ln_sigma=sigma, seed=data_seed, nthreads=16) if levels is None else hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=16)
    ref = None
    for form in (1, 2, 0):
        ch = hammlet_amd.Chain(device=0, seed=1)
        ch.load(x)
        ch.set_model(K, ch.autoprior(0.2, 0.9))
        ch.sample_prior()
        ch.iterate("F", 64, 0)
        ch.iterate("F", 1000 if levels is not None else 100, 10 if levels is not None else 1)   # 100 recorded sweeps
        ch.sync()
        rec = ch.recorded_sweeps()
        pool.set_form(form)
        t0 = time.perf_counter()
        pool.marginals(ch)
        t1 = time.perf_counter()
        info, last = pool.info(), pool.last()
        seg, cnt = ch.marginals_rle()
        if ref is None:
            ref = (seg, cnt)
        same = np.array_equal(seg, ref[0]) and np.array_equal(cnt, ref[1])
        print("%-20s T=%d K=%d, %d recorded sweeps, %d marginal segments | form %s (%s): %.1f ms in all (export + collective + install, 1 rank), collective %.2f ms, %.3f GB on the wire per rank"
              " | pooled marginals equal the dense form's: %s" % (wl, T, K, rec, len(seg), {0: "auto", 1: "dense", 2: "lists"}[form], last["form"], 1e3 * (t1 - t0),
                                                                 info["last_allreduce_ms"], info["last_bytes"] / 1e9, same), flush=True)
        ch.close()
    del x
pool.close()
