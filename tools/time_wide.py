"""More than 16 states (hml_k_wide.h): sweep times of the default path against the reference-compatible mode on config 3's trace
(10^8 positions, five levels; K states in the model), with the per-family brackets of the default path.
    python tools/time_wide.py [K ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammlet_amd as h

T = int(os.environ.get("HML_TIME_T", "100000000"))
x = h.synth_gauss(T, 5, [-2, -1, 0, 1, 2], 0.3, 5000.0, 3, nthreads=16)
for K in [int(a) for a in sys.argv[1:]] or [20, 40, 64]:
    res = {}
    burn, n_timed = int(os.environ.get("HML_TIME_BURN", "60")), int(os.environ.get("HML_TIME_N", "100"))
    modes = (("default", burn, n_timed),) if os.environ.get("HML_TIME_NO_COMPAT") else (("default", burn, n_timed), ("compat", 12, 12))
    for mode, burn, n in modes:
        c = h.Chain(device=0, seed=1)
        if mode == "compat":
            c.set_option("compat", 1)
        c.load(x)
        c.set_model(K, c.autoprior(0.2, 0.9))
        c.sample_prior()
        c.set_recording(marginals=False)
        c.iterate("F", burn, 0)
        c.sync()
        s0 = c.stats()
        t0 = time.perf_counter()
        c.iterate("F", n, 0)
        c.sync()
        dt = time.perf_counter() - t0
        s1 = c.stats()
        blocks = (s1["block_updates"] - s0["block_updates"]) / n
        fam = {}
        if mode == "default":
            c.profile_enable(2)
            c.iterate("F", 20, 0)
            c.sync()
            c.profile_enable(0)
            for nm in ("blocks_compact", "blocks_scatter", "block_stats", "stats_emission", "emission", "forward", "backward_maps", "backward_chain", "counts", "params"):
                ms, cnt = c.profile_get(nm)
                if cnt:
                    fam[nm] = round(1e3 * ms / cnt - 5.3, 1)
        if mode == "default":
            fam["warmup"] = (s0["forward_warmup"], s1["forward_warmup"])
        res[mode] = (1e3 * dt / n, blocks, s1["forward_refits"] - s0["forward_refits"], fam)
        c.close()
    d = res["default"]
    cp = res.get("compat", (float("nan"), float("nan")))
    print("K=%d T=%d: default %.3f ms/sweep (%.0f blocks, %.3g block-updates/s, %d chunks run again) %s | compat %.3f ms/sweep (%.0f blocks, %.3g/s) | %.1fx"
          % (K, T, d[0], d[1], d[1] / d[0] * 1e3, d[2], d[3], cp[0], cp[1], cp[1] / cp[0] * 1e3, (d[1] / d[0]) / (cp[1] / cp[0])), flush=True)
