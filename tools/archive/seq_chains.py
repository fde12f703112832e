import os, sys, time
sys.path.insert(0, "/root/repo")
import bench, hammlet_amd
T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS["c3_1e8_k5_dynamic"]
x = hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=8)
for rep in range(4):
    ch = hammlet_amd.Chain(device=0, seed=1)
    ch.load(x)
    ch.set_model(K, ch.autoprior(0.2, 0.9))
    ch.sample_prior()
    ch.set_recording(marginals=False)
    ch.iterate("F", 100, 0); ch.sync()
    s0 = ch.stats()
    t0 = time.perf_counter(); ch.iterate("F", 1000, 0); ch.sync(); dt = time.perf_counter() - t0
    s1 = ch.stats()
    print("chain %d: %.2f us/sweep, fused_fallbacks %d, W %d" % (rep, 1e6 * dt / 1000, s1["fused_fallbacks"], s1["forward_warmup"]), flush=True)
    if rep != 2: ch.close()
