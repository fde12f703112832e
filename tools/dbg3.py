import sys, numpy as np
sys.path.insert(0, '.')
import hammlet_amd as hml
T, K = 10_000_000, 10
x = hml.synth_gauss(T, K, [v - 4.5 for v in range(10)], 0.3, 5000.0, 4)
c = hml.Chain(seed=1); c.load(x); c.set_model(K, c.autoprior()); c.sample_prior()
prev = c.stats()
for i in range(40):
    c.iterate("F", 1, 0); c.sync()
    st = c.stats()
    th = c.theta()
    print(i, "B", c.num_blocks(), "refits", st["forward_refits"] - prev["forward_refits"], "serial", st["forward_serial"] - prev["forward_serial"],
          "W", st["forward_warmup"], "fallbacks", st["uniform_fallbacks"] - prev["uniform_fallbacks"], "means", np.round(th[0::2], 2), "minvar %.3g" % th[1::2].min())
    prev = st
