"""Stage stamps of the parameter kernel (HML_PARAMS_DEBUG): python tools/r4_params_dbg.py [workload] [sweeps]"""
import os, sys
os.environ["HML_PARAMS_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hammlet_amd
wl = sys.argv[1] if len(sys.argv) > 1 else "c4_1e8_k10"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T, K, levels, sigma, dwell, data_seed = bench.WORKLOADS[wl]
x = hammlet_amd.synth_gauss(T, K, levels, sigma, dwell, data_seed, nthreads=8)
ch = hammlet_amd.Chain(device=0, seed=1, chain_id=0)
ch.load(x)
ch.set_model(K, ch.autoprior(0.2, 0.9))
ch.sample_prior()
ch.set_recording(marginals=False)
for i in range(6):
    ch.iterate("F", n, 0); ch.sync()
