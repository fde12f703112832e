#!/usr/bin/env python3
"""Generates the golden output files under tests/golden/ by running the UNMODIFIED reference binary
(oracle/_ref/hammlet, built by oracle/Makefile from /root/reference/src/main.cpp) on synthetic traces.

Only runs where /root/reference exists (the build container).  The committed fixtures are data: the
reference's output files for inputs that the tests regenerate from (T, K, data seed) with the
repository's own deterministic generator (values are printed with %.9g, which round-trips float32
through the reference's `istream >> float`).

    python tests/golden/make_golden.py
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

REF = os.path.join(REPO, "oracle", "_ref", "hammlet")

# name -> (T, K, data_seed, flags, outputs)
CASES = {
    "c1_fb": (100000, 3, 1, "-s 3 -R 1 -i F 100 1", ["marginals", "sequences", "parameters", "blocks", "compression", "segments"]),
    "c1_default_scheme": (100000, 3, 11, "-s 3 -R 11", ["marginals"]),
    "k4_mixed_scheme": (20000, 4, 3, "-s 4 -R 3 -i M 50 5 D F 60 2 P M 10 1 S F 30 1",
                        ["marginals", "sequences", "parameters", "compression", "segments"]),
    "k3_no_self_transitions": (100000, 3, 1, "-s 3 -R 5 -S -i F 50 1", ["marginals", "parameters"]),
    "k3_priors_multiplier": (100000, 3, 1, "-s 3 -R 7 -m 1.5 -t 1 10 -I 2 -e normal 0.1 0.8 -i M 20 1 F 50 1",
                             ["marginals", "parameters", "compression"]),
    "k6_static_then_dynamic": (100000, 3, 1, "-s 6 -R 8 -t 0.1 -i S F 50 1 P D F 20 1", ["marginals", "parameters"]),
    "k2_thinning": (100000, 3, 1, "-s 2 -R 9 -i F 30 7 M 5 0 F 10 3", ["marginals", "sequences"]),
    "k5_200k": (200000, 5, 7, "-s 5 -R 42 -i F 25 5", ["marginals", "parameters", "compression"]),
    # sizes around the structural edges: powers of two (the `R < size` rule of HaarBreakpointWeights forces
    # breakpoints at T/2, 3T/4, ...), the 65535-position integral-array cell, tiny inputs
    "t16": (16, 2, 21, "-s 2 -R 1 -i F 20 1", ["marginals", "sequences", "blocks", "parameters"]),
    "t1000": (1000, 3, 22, "-s 3 -R 2 -i M 5 1 F 20 1", ["marginals", "sequences", "blocks", "parameters"]),
    # `-O segments` (Records.hpp:208-209: #marginal segments and the length of StateMarginals' count queue before the
    # sweep's last run is added) - with the marginals (c1_fb, k4_mixed_scheme, k40_mixed_scheme, mv_c22 above) and
    # WITHOUT them, where the queue never grows
    "t1000_segments_only": (1000, 3, 22, "-s 3 -R 2 -i M 5 1 F 20 1", ["segments", "sequences"]),
    "t4096": (4096, 3, 23, "-s 3 -R 3 -i F 20 1", ["marginals", "blocks", "parameters"]),
    "t65535": (65535, 3, 24, "-s 3 -R 4 -i F 20 2", ["marginals", "blocks", "parameters"]),
    "t65536": (65536, 3, 25, "-s 3 -R 5 -i F 20 2", ["marginals", "blocks", "parameters"]),
    "t65537": (65537, 3, 26, "-s 3 -R 6 -i F 20 2", ["marginals", "blocks", "parameters"]),
    "t131071": (131071, 4, 27, "-s 4 -R 7 -i F 20 4", ["marginals", "parameters", "compression"]),
    # multivariate / shared parameters ("-s C P D": K = P^D states, D interleaved data dimensions).
    # Trace: dimension d is the univariate generator with data seed + d; the values of a position follow each other.
    "mv_c22": (30000, 2, 51, "-s C 2 2 -R 5 -i F 50 1", ["marginals", "sequences", "parameters", "blocks", "compression", "segments"]),
    "mv_c32_mixed": (30000, 3, 52, "-s C 3 2 -R 6 -i M 20 1 S P F 30 2 D F 10 1", ["marginals", "sequences", "parameters", "compression"]),
    "mv_c23": (20000, 2, 53, "-s C 2 3 -R 7 -i F 30 1", ["marginals", "parameters", "blocks"]),
    "mv_c42_no_self": (70000, 4, 54, "-s C 4 2 -R 8 -S -m 1.3 -i F 20 2", ["marginals", "parameters", "compression"]),
    # more than 16 states (round 4: the reference takes any -s K, main.cpp:112-136; the GPU library runs such models in its
    # reference-compatible mode, hml_k_compat.h)
    "k20_many_states": (60000, 6, 61, "-s 20 -R 12 -i F 30 1", ["marginals", "sequences", "parameters", "blocks"]),
    "k40_mixed_scheme": (30000, 5, 62, "-s 40 -R 13 -t 0.2 2 -i M 10 1 S P F 15 2 D F 5 1", ["marginals", "parameters", "compression", "segments"]),
    "mv_c52_25_states": (40000, 5, 63, "-s C 5 2 -R 14 -i F 20 1", ["marginals", "parameters", "compression"]),
    "k64_most_states": (20000, 4, 64, "-s 64 -R 15 -i F 12 1", ["marginals", "parameters"]),
}


def case_dims(flags):
    """data dimensions of a case: the third token of `-s C P D` (1 for `-s K`)"""
    t = flags.split()
    i = t.index("-s")
    return int(t[i + 3]) if t[i + 1] in ("C", "combinations") and i + 3 < len(t) and t[i + 3].isdigit() else 1


def case_trace(T, K, dseed, flags):
    D = case_dims(flags)
    if D == 1:
        return ol.trace(T, K, dseed)
    return np.stack([ol.trace(T, K, dseed + d) for d in range(D)], axis=1).reshape(-1)


def main():
    if not os.path.exists(REF):
        raise SystemExit("reference binary missing: run `make -C oracle ref` in the build container")
    manifest = {}
    for name, (T, K, dseed, flags, outs) in CASES.items():
        # the data K (levels of the trace) is the first case field; `-s` in flags is the model's K
        x = case_trace(T, K, dseed, flags)
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
        manifest[name] = {"T": T, "trace_levels": K, "data_seed": dseed, "flags": flags, "outputs": outs, "dims": case_dims(flags)}
        print(name, "ok")
    # `-g` prints the parser state (Parser::print, reference src/Parser.hpp:242-269): golden for the host-side parser
    with tempfile.TemporaryDirectory() as tmp:
        inp = os.path.join(tmp, "tiny.txt")
        with open(inp, "w") as f:
            f.write(" ".join(str(v) for v in list(range(1, 17)) + [1, 2, 3, 4]) + "\n")
        cmd = [REF, "-g", "-a", "-s", "4", "-R", "1", "-i", "F", "1", "0", "-f", "tiny.txt", "-o", "out-", ".csv", "-w",
               "-O", "marginals", "blocks", "-t", "0.3", "0.7"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
        with open(os.path.join(HERE, "cli_g_output.txt"), "w") as g:
            g.write("\n".join(r.stdout.splitlines()[:16]) + "\n")
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
