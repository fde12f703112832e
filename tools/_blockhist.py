import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hammlet_amd as h
T = 100000000
x = h.synth_gauss(T, 5, [-2, -1, 0, 1, 2], 0.3, 5000.0, 3, nthreads=16)
for K in (20, 64):
    c = h.Chain(device=0, seed=1)
    c.load(x); c.set_model(K, c.autoprior(0.2, 0.9)); c.sample_prior(); c.set_recording(marginals=False)
    c.iterate("F", 80, 0); c.sync()
    st = np.asarray(c.blocks()).astype(np.int64)
    n = np.diff(np.concatenate([st, [T]])) if st[-1] != T else np.diff(st)
    print("K=%d blocks %d mean %.1f" % (K, len(n), n.mean()), {t: float((n >= t).mean()) for t in (2, 4, 16, 64, 256, 1024, 2048, 4096, 16384)}, "max", int(n.max()), flush=True)
    c.close()
