"""Dense regime timing: the config-3 trace with the breakpoint weights multiplied by 1e9 (every position its own block,
B = T = 10^8; SURVEY 8d "C3u") or the config-5 depth trace; N sweeps after a short burn-in.
usage: python tools/time_dense.py [c3u|c5] [sweeps] [T] [burn-in sweeps, default 70: past the chunk-length measurement]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammlet_amd  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3u"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = 5
if which == "c3u":
    T = int(float(sys.argv[3])) if len(sys.argv) > 3 and float(sys.argv[3]) > 0 else 100_000_000
    x = hammlet_amd.synth_gauss(T, K, [-2, -1, 0, 1, 2], 0.3, 5000.0, 3, nthreads=16)
else:
    T = int(float(sys.argv[3])) if len(sys.argv) > 3 and float(sys.argv[3]) > 0 else 250_000_000
    x = hammlet_amd.synth_depth(T, depth=15.0, ln_sigma=0.15, seed=5, nthreads=16)
c = hammlet_amd.Chain(device=0, seed=1)
c.load(x)
if which == "c3u":
    c.scale_weights(1e9)
c.set_model(K, c.autoprior(0.2, 0.9))
c.sample_prior()
c.set_recording(marginals=False)
c.iterate("F", int(sys.argv[4]) if len(sys.argv) > 4 else 70, 0)
c.sync()
s0 = c.stats()
t0 = time.perf_counter()
c.iterate("F", n, 0)
c.sync()
dt = time.perf_counter() - t0
s1 = c.stats()
B = (s1["block_updates"] - s0["block_updates"]) / n
print("%s T=%d: %.3f ms/sweep, %.3e blocks/sweep, %.3e block-updates/s, refits %d serial %d warm-up %d, sweep_frac %.4f" % (
    which, T, 1e3 * dt / n, B, B * n / dt, s1["forward_refits"] - s0["forward_refits"], s1["forward_serial"] - s0["forward_serial"],
    s1["forward_warmup"], (4.0 * T + B * (36 + 8 * K)) / (dt / n) / 8e12))
import zlib  # noqa: E402
import numpy as np  # noqa: E402
print("  check: theta crc %08x, states crc %08x" % (zlib.crc32(np.ascontiguousarray(c.theta()).tobytes()), zlib.crc32(np.ascontiguousarray(c.states()).tobytes())))
