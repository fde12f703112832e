import sys, time
sys.path.insert(0, "/root/repo")
import hammlet_amd as h
for K in (8, 12, 16):
    T = 20_000_000
    x = h.synth_gauss(T, K, [i - (K - 1) / 2 for i in range(K)], 0.3, 5000.0, 3, nthreads=16)
    c = h.Chain(device=0, seed=1)
    c.load(x); c.set_model(K, c.autoprior(0.2, 0.9)); c.sample_prior(); c.set_recording(marginals=False)
    c.iterate("F", 300, 0); c.sync()
    s0 = c.stats(); t0 = time.perf_counter(); c.iterate("F", 300, 0); c.sync(); dt = time.perf_counter() - t0; s1 = c.stats()
    c.profile_enable(2); c.iterate("F", 100, 0); c.sync(); c.profile_enable(0)
    fam = {n: round(1e3 * c.profile_get(n)[0] / max(1, c.profile_get(n)[1]) - 5.3, 1) for n in ("blocks_compact", "forward", "backward_maps", "backward_chain", "counts", "params")}
    print("K=%d T=%d: %.1f us/sweep, B %.0f, refits %d" % (K, T, 1e6 * dt / 300, (s1["block_updates"] - s0["block_updates"]) / 300, s1["forward_refits"] - s0["forward_refits"]), fam, flush=True)
    c.close()
