"""Sweeps 5-25 of fresh chains (what `bench.py --steps 20 --warmup 5` times), several chains, environment as given."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hammlet_amd
T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS["c3_1e8_k5_dynamic"]
x = hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=8)
res = []
for rep in range(6):
    ch = hammlet_amd.Chain(device=0, seed=1)
    ch.load(x)
    ch.set_model(K, ch.autoprior(0.2, 0.9))
    ch.sample_prior()
    ch.set_recording(marginals=False)
    ch.iterate("F", 5, 0); ch.sync()
    s0 = ch.stats()
    t0 = time.perf_counter(); ch.iterate("F", 20, 0); ch.sync(); dt = time.perf_counter() - t0
    s1 = ch.stats()
    res.append("%.1f us (W %d, refits %d)" % (1e6 * dt / 20, s1["forward_warmup"], s1["forward_refits"] - s0["forward_refits"]))
    ch.close()
print(os.environ.get("LABEL", ""), " | ".join(res))
