"""First 400 sweeps of young chains (configs 2, 3, 4; two chain seeds each) under a warm-up policy given by the environment
(HML_FWD_BURNIN_SWEEPS, HML_FWD_QUIET): total milliseconds, refits, and the time of sweeps 5..25 (what the driver's
`--steps 20 --warmup 5` measures).   python tools/r4_burnin.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hammlet_amd
for wl in ("c3_1e8_k5_dynamic", "c2_1e7_k5", "c4_1e8_k10"):
    T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS[wl]
    x = hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=16)
    for seed in (1, 2):
        ch = hammlet_amd.Chain(device=0, seed=seed)
        ch.load(x)
        ch.set_model(K, ch.autoprior(0.2, 0.9))
        ch.sample_prior()
        ch.set_recording(marginals=False)
        ch.iterate("F", 5, 0); ch.sync()
        t0 = time.perf_counter(); ch.iterate("F", 20, 0); ch.sync(); t1 = time.perf_counter()
        ch.iterate("F", 375, 0); ch.sync(); t2 = time.perf_counter()
        s = ch.stats()
        print("%-18s seed %d: sweeps 5-25 %.4f ms/sweep | sweeps 5-400 %.2f ms | refits %d serial %d | warm-up now %d" % (
            wl, seed, 1e3 * (t1 - t0) / 20, 1e3 * (t2 - t0), s["forward_refits"], s["forward_serial"], s["forward_warmup"]))
        ch.close()
