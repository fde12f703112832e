import sys, time, numpy as np
sys.path.insert(0, '.')
import hammlet_amd as hml
T = 10_000_000
x = hml.synth_gauss(T, 5, [-2, -1, 0, 1, 2], 0.3, 5000.0, 2)
for K in (5, 8, 12):
    c = hml.Chain(seed=1); c.load(x); c.set_model(K, c.autoprior()); c.sample_prior()
    c.iterate("F", 100, 0); c.sync()
    s0 = c.stats(); t0 = time.perf_counter()
    c.iterate("F", 200, 0); c.sync()
    t1 = time.perf_counter(); s1 = c.stats()
    print("K", K, "ms/sweep %.3f" % ((t1 - t0) / 200 * 1e3), "blocks/sweep", (s1["block_updates"] - s0["block_updates"]) / 200,
          "refits/sweep", (s1["forward_refits"] - s0["forward_refits"]) / 200, "serial/sweep", (s1["forward_serial"] - s0["forward_serial"]) / 200,
          "W", s1["forward_warmup"], "means", np.round(np.sort(c.theta()[0::2]), 2))
    c.close()
