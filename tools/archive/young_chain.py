"""How fast are the first sweeps of a chain?  ms per sweep in groups of 5 sweeps (a sync between groups), config 3.
    python tools/young_chain.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hammlet_amd
T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS["c3_1e8_k5_dynamic"]
x = hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=8)
for rep in range(2):
    ch = hammlet_amd.Chain(device=0, seed=1 + rep)
    ch.load(x)
    ch.set_model(K, ch.autoprior(0.2, 0.9))
    ch.sample_prior()
    ch.set_recording(marginals=False)
    ch.sync()
    out = []
    for g in range(12):
        s0 = ch.stats()
        t0 = time.perf_counter(); ch.iterate("F", 5, 0); ch.sync(); dt = time.perf_counter() - t0
        s1 = ch.stats()
        out.append("%d-%d: %.1f us (W %d, refits %d, B %d)" % (5 * g, 5 * g + 5, 1e6 * dt / 5, s1["forward_warmup"], s1["forward_refits"] - s0["forward_refits"], (s1["block_updates"] - s0["block_updates"]) // 5))
    print("chain %d: " % rep + " | ".join(out), flush=True)
    ch.close()
