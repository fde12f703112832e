#!/usr/bin/env python3
"""Golden data for the statistical bridge between the GPU chain (Philox streams, design deviation D1) and the
reference (one sequential mt19937).  The UNMODIFIED reference binary (oracle/_ref/hammlet) is run with many seeds
on each input; committed are
  * the marginals and parameters files of two of those runs (`marginals_seed<S>.csv`, `parameters_seed<S>.csv`),
  * for every run its posterior-mean parameters, states sorted by mean (`bridge_manifest.json: reference_runs`).
The GPU test (tests/test_gpu_reference_bridge.py) compares its chains with the two full runs and takes the spread
among the reference runs as the yardstick; on the 5-state input the sampler has several posterior modes (which one
a chain ends in depends on its first prior draw), and the test also compares HOW OFTEN chains reach the main mode.
Only runs in the build container.

    python tests/golden/make_bridge_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from tests import oracle_lib as ol  # noqa: E402
from tests import bridge_util as bu  # noqa: E402

REF = os.path.join(REPO, "oracle", "_ref", "hammlet")

# name -> (T, trace (levels, sigma, mean dwell), data seed, model flags without -R, scheme, reference seeds; the files
# of the first two seeds that reach the main mode are kept).
# bridge_c1 is config 1 with 100 burn-in sweeps.  The 5-state case uses short dwell times (300): on the
# 5000-dwell trace of tests/golden/k5_200k two REFERENCE seeds end in different posterior modes (21 % of the
# positions differ in their arg-max state, one chain splits a level in two) and none of them is the main one.
CASES = {
    "bridge_c1": (100000, ([-1, 0, 1], 0.2, 2000), 1, "-s 3", "-i F 100 0 F 300 1", list(range(1, 13))),
    "bridge_k5": (200000, ([-2, -1, 0, 1, 2], 0.2, 300), 7, "-s 5", "-i F 100 0 F 300 2", list(range(42, 72))),
}


def main():
    if not os.path.exists(REF):
        raise SystemExit("reference binary missing: run `make -C oracle ref` in the build container")
    manifest = {}
    for name, (T, (levels, sigma, dwell), dseed, flags, scheme, seeds) in CASES.items():
        K = int(flags.split()[1])
        x = ol.synth_gauss(T, len(levels), levels, sigma, dwell, dseed)
        d = os.path.join(HERE, name)
        os.makedirs(d, exist_ok=True)
        runs, kept = [], []
        with tempfile.TemporaryDirectory() as tmp:
            inp = os.path.join(tmp, "in.txt")
            np.savetxt(inp, x, fmt="%.9g")
            for s in seeds:
                cmd = [REF, "-f", inp, "-o", os.path.join(tmp, "ref-"), ".csv", "-w", "-a"] + flags.split() + ["-R", str(s)] + \
                      scheme.split() + ["-O", "marginals", "parameters"]
                r = subprocess.run(cmd, capture_output=True, text=True)
                assert r.returncode == 0, r.stderr
                texts = {o: open(os.path.join(tmp, "ref-%s.csv" % o)).read() for o in ("marginals", "parameters")}
                par = bu.parse_parameters(texts["parameters"], K).mean(axis=0)
                order = np.argsort(par[:, 0], kind="stable")
                mean, var = par[order, 0], par[order, 1]
                main_mode = bool(np.abs(mean - np.asarray(levels, float)).max() < bu.MAIN_MODE_TOL)
                runs.append({"seed": s, "mean": [float(v) for v in mean], "var": [float(v) for v in var], "main_mode": main_mode})
                if main_mode and len(kept) < 2:
                    kept.append(s)
                    for o, t in texts.items():
                        with open(os.path.join(d, "%s_seed%d.csv" % (o, s)), "w") as g:
                            g.write(t)
        manifest[name] = {"T": T, "levels": levels, "sigma": sigma, "dwell": dwell, "data_seed": dseed, "flags": flags, "scheme": scheme,
                          "seeds": kept, "reference_runs": runs}
        print(name, "ok: %d of %d reference runs in the main mode" % (sum(r["main_mode"] for r in runs), len(runs)))
    with open(os.path.join(HERE, "bridge_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
