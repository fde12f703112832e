"""Weakly compressed sweeps with the scheme `F n 10` (every tenth sweep recorded, marginals on) against plain sweeps:
    python tools/time_dense_recorded.py [c3u|c5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammlet_amd
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
K = 5
if which == "c3u":
    T = 100_000_000
    x = hammlet_amd.synth_gauss(T, K, [-2, -1, 0, 1, 2], 0.3, 5000.0, 3, nthreads=16)
else:
    T = 250_000_000
    x = hammlet_amd.synth_depth(T, depth=15.0, ln_sigma=0.15, seed=5, nthreads=16)
c = hammlet_amd.Chain(device=0, seed=1)
c.load(x)
if which == "c3u":
    c.scale_weights(1e9)
c.set_model(K, c.autoprior(0.2, 0.9))
c.sample_prior()
c.set_recording(marginals=False)
c.iterate("F", 70, 0); c.sync()
for label, rec, thin in (("plain", False, 0), ("F n 10, marginals", True, 10), ("F n 1, marginals", True, 1)):
    c.set_recording(marginals=rec)
    t0 = time.perf_counter(); c.iterate("F", 20, thin); c.sync(); dt = time.perf_counter() - t0
    print("%s %-20s %.3f ms/sweep" % (which, label, 1e3 * dt / 20), flush=True)
t0 = time.perf_counter(); seg, cnt = c.marginals_rle(); dt = time.perf_counter() - t0
print("%s marginals_rle: %d segments in %.3f s" % (which, len(seg), dt))
