"""Time plain sweeps (no event brackets) of the bench workload: python tools/time_sweeps.py [workload] [sweeps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hammlet_amd
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_1e8_k5_dynamic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS[wl]
x = hammlet_amd.synth_depth(T, depth=dwell, ln_sigma=sigma, seed=data_seed, nthreads=8) if levels is None else hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=8)
ch = hammlet_amd.Chain(device=0, seed=1)
ch.load(x)
ch.set_model(K, ch.autoprior(0.2, 0.9))
ch.sample_prior()
ch.set_recording(marginals=False)
ch.iterate("F", 40, 0); ch.sync()
s0 = ch.stats()
t0 = time.perf_counter(); ch.iterate("F", n, 0); ch.sync(); t1 = time.perf_counter()
s1 = ch.stats()
print("%s: %.4f ms/sweep, %.3e block-updates/s, refits %d serial %d, warm-up %d -> %d, B %d" % (
      wl, 1e3 * (t1 - t0) / n, (s1["block_updates"] - s0["block_updates"]) / (t1 - t0),
      s1["forward_refits"] - s0["forward_refits"], s1["forward_serial"] - s0["forward_serial"], s0["forward_warmup"], s1["forward_warmup"],
      (s1["block_updates"] - s0["block_updates"]) // n))
