"""Weakly compressed sweeps for several numbers of states: the config-3 trace shape with every position its own block
(weights x 1e9), T = 5e7.   python tools/time_dense_states.py [K ...]"""
import sys, time
sys.path.insert(0, "/root/repo")
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammlet_amd as h
Ks = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 6, 8, 10, 16]
T = 50_000_000
for K in Ks:
    x = h.synth_gauss(T, K, [i - (K - 1) / 2 for i in range(K)], 0.3, 5000.0, 3, nthreads=16)
    c = h.Chain(device=0, seed=1)
    c.load(x); c.scale_weights(1e9)
    c.set_model(K, c.autoprior(0.2, 0.9)); c.sample_prior(); c.set_recording(marginals=False)
    c.iterate("F", 70, 0); c.sync()
    s0 = c.stats(); t0 = time.perf_counter(); c.iterate("F", 20, 0); c.sync(); dt = time.perf_counter() - t0; s1 = c.stats()
    B = (s1["block_updates"] - s0["block_updates"]) / 20
    c.profile_enable(2); c.iterate("F", 5, 0); c.sync(); c.profile_enable(0)
    fam = {}
    for n in ("blocks_compact", "blocks_scatter", "trellis", "trellis_repair", "backward_chain", "counts"):
        ms, cnt = c.profile_get(n)
        if cnt: fam[n] = round(ms / cnt, 3)
    print("K=%d T=%d: %.3f ms/sweep, B %.3e, %.3e block-updates/s, sweep_frac %.3f, refits/sweep %.0f, W %d | %s" % (
        K, T, 1e3 * dt / 20, B, B * 20 / dt, (4.0 * T + B * (36 + 8 * K)) / (dt / 20) / 8e12,
        (s1["forward_refits"] - s0["forward_refits"]) / 20, s1["forward_warmup"], fam), flush=True)
    c.close()
