"""Duration of the fused block kernel against the trace length (no 2x step when the grid outgrows the
resident slots): Gaussian generator of config 3 at several T, HIP-event brackets around every launch of 300 sweeps.
usage: python tools/time_fused_T.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammlet_amd  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
TS = [int(float(v)) for v in sys.argv[2:]] or [25_000_000, 50_000_000, 100_000_000, 101_000_000, 150_000_000, 200_000_000, 250_000_000, 400_000_000]
levels = [-2, -1, 0, 1, 2] if K == 5 else [x - (K - 1) / 2 for x in range(K)]
print("T, blocks/sweep, fused kernel us (bracket - empty bracket), sweep us, fused_fallbacks")
for T in TS:
    x = hammlet_amd.synth_gauss(T, K, levels, 0.3, 5000.0, 3, nthreads=16)
    c = hammlet_amd.Chain(device=0, seed=1)
    c.load(x)
    c.set_model(K, c.autoprior(0.2, 0.9))
    c.sample_prior()
    c.set_recording(marginals=False)
    c.iterate("F", 300, 0)
    c.sync()
    s0 = c.stats()
    t0 = time.perf_counter()
    c.iterate("F", 1000, 0)
    c.sync()
    dt = time.perf_counter() - t0
    c.profile_enable(2)
    c.iterate("F", 300, 0)
    c.sync()
    c.profile_enable(0)
    ms, n = c.profile_get("blocks_compact")
    # an empty bracket measures ~5 us on this stack (bench.py reports it per run)
    s1 = c.stats()
    print("%d, %.0f, %.2f, %.2f, %d" % (T, (s1["block_updates"] - s0["block_updates"]) / 1300.0, 1e3 * ms / max(n, 1) - 5.3, 1e6 * dt / 1000, s1["fused_fallbacks"]), flush=True)
    c.close()
    del x
