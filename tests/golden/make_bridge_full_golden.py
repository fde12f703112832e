#!/usr/bin/env python3
"""Statistical bridge at BASELINE config 3's FULL size: the UNMODIFIED reference binary (oracle/_ref/hammlet) with several
seeds on the 10^8-position, 5-level trace of bench.py's headline workload, 100 burn-in sweeps and 100 sweeps of which every
10th is recorded.  Committed under tests/golden/bridge_c3/: the marginals (xz) and parameters files of the first two runs
that reach the main posterior mode, and in bridge_manifest.json the posterior-mean parameters of every run (the yardstick
of tests/test_gpu_reference_bridge.py).  The runs go side by side (one process per seed, ~3 GB each); ~2 minutes.
Only runs in the build container.

    python tests/golden/make_bridge_full_golden.py
"""
import ctypes as C
import json
import lzma
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
NAME = "bridge_c3"
T, LEVELS, SIGMA, DWELL, DSEED = 100_000_000, [-2, -1, 0, 1, 2], 0.3, 5000, 3     # bench.py WORKLOADS["c3_1e8_k5_dynamic"]
FLAGS, SCHEME, SEEDS = "-s 5", "-i F 100 0 F 100 10", [1, 2, 3, 4, 5, 6]


def main():
    if not os.path.exists(REF):
        raise SystemExit("reference binary missing: run `make -C oracle ref` in the build container")
    lib = ol.load()
    lib.orc_write_text.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_int]
    K = int(FLAGS.split()[1])
    x = ol.synth_gauss(T, len(LEVELS), LEVELS, SIGMA, DWELL, DSEED)
    d = os.path.join(HERE, NAME)
    os.makedirs(d, exist_ok=True)
    runs, kept = [], []
    with tempfile.TemporaryDirectory() as tmp:
        inp = os.path.join(tmp, "in.txt")
        assert lib.orc_write_text(x.ctypes.data, x.size, inp.encode(), 8) == 0
        del x
        procs = []
        for s in SEEDS:
            cmd = [REF, "-f", inp, "-o", os.path.join(tmp, "ref%d-" % s), ".csv", "-w", "-a"] + FLAGS.split() + ["-R", str(s)] + \
                  SCHEME.split() + ["-O", "marginals", "parameters"]
            procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
        for s, p in zip(SEEDS, procs):
            _, err = p.communicate()
            assert p.returncode == 0, err
        for s in SEEDS:
            texts = {o: open(os.path.join(tmp, "ref%d-%s.csv" % (s, o))).read() for o in ("marginals", "parameters")}
            par = bu.parse_parameters(texts["parameters"], K).mean(axis=0)
            order = np.argsort(par[:, 0], kind="stable")
            mean, var = par[order, 0], par[order, 1]
            main_mode = bool(np.abs(mean - np.asarray(LEVELS, float)).max() < bu.MAIN_MODE_TOL)
            runs.append({"seed": s, "mean": [float(v) for v in mean], "var": [float(v) for v in var], "main_mode": main_mode,
                         "marginal_segments": texts["marginals"].count("\n")})
            if main_mode and len(kept) < 2:
                kept.append(s)
                with open(os.path.join(d, "marginals_seed%d.csv.xz" % s), "wb") as g:
                    g.write(lzma.compress(texts["marginals"].encode(), preset=9 | lzma.PRESET_EXTREME))
                with open(os.path.join(d, "parameters_seed%d.csv" % s), "w") as g:
                    g.write(texts["parameters"])
    mpath = os.path.join(HERE, "bridge_manifest.json")
    with open(mpath) as f:
        manifest = json.load(f)
    manifest[NAME] = {"T": T, "levels": LEVELS, "sigma": SIGMA, "dwell": DWELL, "data_seed": DSEED, "flags": FLAGS, "scheme": SCHEME,
                      "seeds": kept, "reference_runs": runs}
    print(NAME, "ok: %d of %d reference runs in the main mode" % (sum(r["main_mode"] for r in runs), len(runs)))
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
