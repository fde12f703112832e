"""Sweep time of a multivariate chain: python tools/time_mv.py [P=2] [D=2] [T=10000000] [sweeps=500]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hammlet_amd
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
n = int(sys.argv[4]) if len(sys.argv) > 4 else 500
levels = [float(i) - (P - 1) / 2 for i in range(P)]
x = np.stack([hammlet_amd.synth_gauss(T, P, levels, 0.3, 5000.0, 11 + d, nthreads=8) for d in range(D)], axis=1).reshape(-1)
ch = hammlet_amd.Chain(device=0, seed=1)
ch.set_dimensions(D, P)
ch.load(x)
ch.set_model(P ** D, ch.autoprior(0.2, 0.9))
ch.sample_prior()
ch.set_recording(marginals=False)
ch.iterate("F", 60, 0); ch.sync()
s0 = ch.stats()
t0 = time.perf_counter(); ch.iterate("F", n, 0); ch.sync(); t1 = time.perf_counter()
s1 = ch.stats()
print("C %d %d, T=%d: %.4f ms/sweep, %.3e block-updates/s, B %d, refits %d" % (P, D, T, 1e3 * (t1 - t0) / n,
      (s1["block_updates"] - s0["block_updates"]) / (t1 - t0), (s1["block_updates"] - s0["block_updates"]) // n, s1["forward_refits"] - s0["forward_refits"]))
