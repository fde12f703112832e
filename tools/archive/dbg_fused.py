import os, sys
sys.path.insert(0, os.getcwd())
import hammlet_amd
T, K = 100_000_000, 5
x = hammlet_amd.synth_gauss(T, K, [-2, -1, 0, 1, 2], 0.3, 5000.0, 3, nthreads=16)
c = hammlet_amd.Chain(device=0, seed=1)
c.load(x)
c.set_model(K, c.autoprior(0.2, 0.9))
c.sample_prior()
c.set_recording(marginals=False)
c.iterate("F", 300, 0)
c.sync()
c.iterate("F", 1, 0)
c.sync()
c.iterate("F", 1, 0)
c.sync()
