"""Wall clock of the first sweeps of a chain, ten at a time (burn-in: repairs, adapting warm-up, changing compression):
python tools/burnin_profile.py [workload] [groups]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hammlet_amd
wl = sys.argv[1] if len(sys.argv) > 1 else "c5_2.5e8_depth_k5"
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS[wl]
x = hammlet_amd.synth_depth(T, depth=dwell, ln_sigma=sigma, seed=data_seed, nthreads=8) if levels is None else hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=8)
ch = hammlet_amd.Chain(device=0, seed=1)
ch.load(x)
ch.set_model(K, ch.autoprior(0.2, 0.9))
ch.sample_prior()
ch.set_recording(marginals=False)
prev = ch.stats()
for g in range(groups):
    n = int(os.environ.get("PER", "10"))
    t0 = time.perf_counter(); ch.iterate("F", n, 0); ch.sync(); t1 = time.perf_counter()
    s = ch.stats()
    print("sweeps %3d-%3d: %8.4f ms/sweep  B %10d  refits %8d  serial %6d  warm-up %d" % (
        n * g, n * g + n - 1, 1e3 * (t1 - t0) / n, (s["block_updates"] - prev["block_updates"]) // n,
        s["forward_refits"] - prev["forward_refits"], s["forward_serial"] - prev["forward_serial"], s["forward_warmup"]))
    prev = s
