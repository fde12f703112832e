import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hammlet_amd as hml
from tests import oracle_lib as ol, bridge_util as bu
from tests.test_gpu_reference_bridge import _posterior_signal, _run_scheme
gold = os.path.join(ROOT, "tests", "golden")
for case in ("k20_many_states", "k40_mixed_scheme"):
    m = json.load(open(os.path.join(gold, "manifest.json")))[case]
    fl = m["flags"].split(); K = int(fl[fl.index("-s") + 1])
    t_off, t_diag = (float(fl[fl.index("-t") + 1]), float(fl[fl.index("-t") + 2])) if "-t" in fl else (0.5, 0.5)
    toks, sch, i = [], fl[fl.index("-i") + 1:], 0
    while i < len(sch):
        if sch[i] in ("P", "S", "D"): toks.append(sch[i]); i += 1
        else: toks.append((sch[i], int(sch[i + 1]), int(sch[i + 2]))); i += 3
    L = m["trace_levels"]
    x, ts = hml.synth_gauss(m["T"], L, ol.LEVELS[L], ol.SIGMA[L], ol.DWELL[L], m["data_seed"], with_states=True)
    truth = np.asarray(ol.LEVELS[L], np.float64)[ts]
    seg, cnt = bu.parse_marginals(open(os.path.join(gold, case, "marginals.csv")).read(), K)
    ref = _posterior_signal(seg, cnt, bu.parse_parameters(open(os.path.join(gold, case, "parameters.csv")).read(), K))
    rmse = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
    sig = []
    for mode in ("default", "compat"):
        errs = []
        for seed in range(1, 13):
            g = hml.Chain(device=0, seed=seed)
            if mode == "compat": g.set_option("compat", 1)
            g.load(x); g.set_model(K, g.autoprior(0.2, 0.9), t_off, t_diag)
            rows = []
            g.set_recording(marginals=True, callback=lambda ch, i: rows.append(ch.theta().astype(np.float64)))
            _run_scheme(g, toks)
            gs, gc = g.marginals_rle(); gc = np.pad(gc, ((0, 0), (0, K - gc.shape[1])))
            s = _posterior_signal(np.asarray(gs, np.int64), np.asarray(gc, np.int64), np.asarray(rows).reshape(len(rows), K, 2))
            errs.append((round(rmse(s, truth), 4), round(rmse(s, ref), 4)))
            g.close()
        print(case, mode, "ref err", round(rmse(ref, truth), 4), "chains (err vs truth, vs ref):", errs, flush=True)
