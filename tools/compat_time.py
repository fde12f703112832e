"""Time of a sweep of the reference-compatible mode (option "compat"): python tools/compat_time.py [workload] [sweeps]   (BURNIN=n sweeps first)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hammlet_amd
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_1e8_k5_dynamic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS[wl]
x = hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=8)
ch = hammlet_amd.Chain(device=0, seed=1, chain_id=0)
ch.set_option("compat", 1)
ch.load(x)
ch.set_model(K, ch.autoprior(0.2, 0.9))
ch.sample_prior()
ch.set_recording(marginals=False)
ch.iterate("F", int(os.environ.get("BURNIN", "2")), 0); ch.sync()
b0 = ch.stats()["block_updates"]
t0 = time.perf_counter()
ch.iterate("F", n, 0); ch.sync()
t1 = time.perf_counter()
b = ch.stats()["block_updates"] - b0
print("[compat] %s: %.2f ms per sweep, %d blocks per sweep, %.3e block-updates/s" % (wl, 1e3 * (t1 - t0) / n, b // n, b / (t1 - t0)))
