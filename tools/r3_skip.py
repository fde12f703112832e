"""development only: time of the first trellis pass for builds with row stages disabled (HML_TR2_SKIP); results are wrong,
errors the chain raises are ignored.  usage: python tools/r3_skip.py L lib1.so lib2.so ...   (one subprocess per library)"""
import os, subprocess, sys
if sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import hammlet_amd
    T, K = 100_000_000, 5
    x = hammlet_amd.synth_gauss(T, K, [-2, -1, 0, 1, 2], 0.3, 5000.0, 3, nthreads=16)
    c = hammlet_amd.Chain(device=0, seed=1)
    c.load(x); c.scale_weights(1e9)
    c.set_model(K, c.autoprior(0.2, 0.9)); c.sample_prior(); c.set_recording(marginals=False)
    c.profile_enable(2)
    try:
        c.iterate("F", 6, 0); c.sync()
    except Exception as e:
        print("   (", str(e)[:60], ")")
    ms, n = c.profile_get("trellis")
    print("%s: trellis %.3f ms (%d launches)" % (os.path.basename(os.environ["HML_LIBRARY"]), ms / max(n, 1), n), flush=True)
    os._exit(0)
for lib in sys.argv[2:]:
    env = dict(os.environ, HML_LIBRARY=os.path.abspath(lib), HML_TRELLIS_L=sys.argv[1], HML_TRELLIS_TUNE="0")
    subprocess.run([sys.executable, __file__, "child"], env=env)
