"""How often do chains of the default path reach the main posterior mode of a bridge case (tests/golden/bridge_manifest.json) - next
to the reference's own rate?   python tools/bridge_probe.py bridge_c3 24"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hammlet_amd as h
from tests import bridge_util as bu

name = sys.argv[1] if len(sys.argv) > 1 else "bridge_c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
c = bu.manifest()[name]
K = int(c["flags"].split()[1])
x = h.synth_gauss(c["T"], len(c["levels"]), c["levels"], c["sigma"], c["dwell"], c["data_seed"], nthreads=16)
toks = c["scheme"].split()[1:]
hits = 0
for seed in range(1, n + 1):
    g = h.Chain(device=0, seed=seed)
    if os.environ.get("PROBE_COMPAT") == "1":
        g.set_option("compat", 1)
    g.load(x)
    g.set_model(K, g.autoprior(0.2, 0.9))
    g.sample_prior()
    rows = []
    g.set_recording(marginals=True, callback=lambda ch, i: rows.append(ch.theta().astype(np.float64)))
    for i in range(0, len(toks), 3):
        g.iterate(toks[i], int(toks[i + 1]), int(toks[i + 2]))
    g.sync()
    par = np.asarray(rows).reshape(len(rows), K, 2)
    pm = np.sort(par.mean(axis=0)[:, 0])
    ok = bool(np.abs(pm - np.asarray(c["levels"], float)).max() < bu.MAIN_MODE_TOL)
    hits += ok
    print(seed, "main" if ok else "    ", np.round(pm, 3), "| first / last recorded:", np.round(np.sort(par[0, :, 0]), 2), np.round(np.sort(par[-1, :, 0]), 2), flush=True)
    g.close()
print("%d of %d chains in the main mode; reference: %d of %d" % (hits, n, sum(r["main_mode"] for r in c["reference_runs"]), len(c["reference_runs"])))
