"""Dense-regime A/B of the first trellis pass: one process (one library, HML_LIBRARY), several (rows, L, warm-up) settings.
usage: python tools/r3_dense.py c3u|c5 T sweeps rows:L[:W[:ckpt]] ...      e.g.  c3u 1e8 20 0:224 1:224 1:512
prints ms/sweep and the per-family event times of the timed sweeps (profile level 2: every launch bracketed, so the sweep
time with brackets is a little above the plain one - both are printed)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hammlet_amd  # noqa: E402

which = sys.argv[1]
T = int(float(sys.argv[2]))
n = int(sys.argv[3])
K = 5
if which == "c3u":
    x = hammlet_amd.synth_gauss(T, K, [-2, -1, 0, 1, 2], 0.3, 5000.0, 3, nthreads=16)
else:
    x = hammlet_amd.synth_depth(T, depth=15.0, ln_sigma=0.15, seed=5, nthreads=16)
ref_states = None
for cfg in sys.argv[4:]:
    parts = cfg.split(":")
    os.environ["HML_TRELLIS_ROWS"] = parts[0]
    os.environ["HML_TRELLIS_L"] = parts[1]
    os.environ["HML_TRELLIS_TUNE"] = "0"
    os.environ["HML_TRELLIS_CKPT"] = parts[3] if len(parts) > 3 else "1"
    if len(parts) > 4 and parts[4]: os.environ["HML_TRE_REFIT_SHIFTS"] = parts[4].replace("/", ",")
    os.environ["HML_STAGE_BITS"] = parts[5] if len(parts) > 5 else "1"
    c = hammlet_amd.Chain(device=0, seed=1)
    c.load(x)
    if which == "c3u":
        c.scale_weights(1e9)
    c.set_model(K, c.autoprior(0.2, 0.9))
    c.sample_prior()
    c.set_recording(marginals=False)
    c.iterate("F", int(os.environ.get("R3_BURN", "30")), 0)
    c.sync()
    s0 = c.stats()
    t0 = time.perf_counter()
    c.iterate("F", n, 0)
    c.sync()
    dt = time.perf_counter() - t0
    s1 = c.stats()
    B = (s1["block_updates"] - s0["block_updates"]) / n
    st = c.states()
    import zlib
    crc = zlib.crc32(st.tobytes())
    c.profile_enable(2)
    c.iterate("F", 5, 0)
    c.sync()
    fam = {}
    for name in ("blocks_compact", "blocks_scatter", "trellis", "trellis_repair", "backward_chain", "counts", "params"):
        try:
            ms, cnt = c.profile_get(name)
            if cnt:
                fam[name] = ms / cnt
        except Exception:
            pass
    c.profile_enable(0)
    print("%s rows=%s L=%s ckpt=%s shifts=%s bits=%s: %.3f ms/sweep, B %.3e, refits/sweep %.0f serial %d W %d, sweep_frac %.4f, crc %08x | %s" % (
        which, parts[0], parts[1], os.environ["HML_TRELLIS_CKPT"], os.environ.get("HML_TRE_REFIT_SHIFTS", "-"), os.environ["HML_STAGE_BITS"], 1e3 * dt / n, B, (s1["forward_refits"] - s0["forward_refits"]) / n, s1["forward_serial"] - s0["forward_serial"],
        s1["forward_warmup"], (4.0 * T + B * (36 + 8 * K)) / (dt / n) / 8e12, crc,
        " ".join("%s %.3f" % (k, v) for k, v in fam.items())), flush=True)
    c.close()
