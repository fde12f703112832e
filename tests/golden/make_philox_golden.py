#!/usr/bin/env python3
"""Philox goldens: output files of the UNMODIFIED reference translation unit compiled with
`-include oracle/shim.hpp`, i.e. with its std::mt19937 replaced by the sequential Philox engine
(oracle/_ref/hammlet_philox, built by oracle/Makefile from /root/reference/src/main.cpp).  The CPU checker in mode
`--rng 1` (the same engine behind the restated libstdc++ distributions' original: libstdc++ itself) must reproduce them
byte for byte (tests/test_oracle_golden.py) - which pins the Philox generator, its key/counter layout and its word order
against the reference's own consumption pattern.  Only runs in the build container.

    python tests/golden/make_philox_golden.py
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

REF = os.path.join(REPO, "oracle", "_ref", "hammlet_philox")

# name -> (T, data K, data seed, flags, outputs)
CASES = {
    "philox_c1_fb": (100000, 3, 1, "-s 3 -R 1 -i F 100 1", ["marginals", "sequences", "parameters", "blocks", "compression"]),
    "philox_c1_default_scheme": (100000, 3, 11, "-s 3 -R 11", ["marginals", "parameters"]),
    "philox_k4_mixed_scheme": (20000, 4, 3, "-s 4 -R 3 -i M 50 5 D F 60 2 P M 10 1 S F 30 1", ["marginals", "sequences", "parameters", "compression"]),
    "philox_k5_no_self": (200000, 5, 7, "-s 5 -R 42 -S -t 1 10 -i F 25 5", ["marginals", "parameters", "compression"]),
}


def main():
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/hammlet_philox missing: run `make -C oracle ref` in the build container")
    manifest = {}
    for name, (T, K, dseed, flags, outs) in CASES.items():
        x = ol.trace(T, K, dseed)
        d = os.path.join(HERE, name)
        os.makedirs(d, exist_ok=True)
        with tempfile.TemporaryDirectory() as tmp:
            inp = os.path.join(tmp, "in.txt")
            np.savetxt(inp, x, fmt="%.9g")
            cmd = [REF, "-f", inp, "-o", os.path.join(tmp, "ref-"), ".csv", "-w", "-a"] + flags.split() + ["-O"] + outs
            r = subprocess.run(cmd, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            for o in outs:
                with open(os.path.join(tmp, "ref-%s.csv" % o)) as f, open(os.path.join(d, o + ".csv"), "w") as g:
                    g.write(f.read())
            with open(os.path.join(d, "stdout.txt"), "w") as g:
                g.write(r.stdout)
        manifest[name] = {"T": T, "trace_levels": K, "data_seed": dseed, "flags": flags, "outputs": outs, "dims": 1}
        print(name, "ok")
    with open(os.path.join(HERE, "philox_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
