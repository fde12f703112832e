#!/usr/bin/env python3
"""Stage-level golden vectors from the UNMODIFIED reference headers: runs oracle/_ref/ref_harness (built by
oracle/Makefile from oracle/ref_harness.cpp + /root/reference/src) on small synthetic traces and commits what every
construction stage produced - maxlet coefficients, noise estimate, breakpoint weights, integral array, auto prior,
and per threshold the block list, block statistics and emission terms - as tests/golden/stages/*.npz.

Only runs where /root/reference exists (the build container).  The fixtures are data; arrays of more than 8192
elements are committed as SHA-256 digests of their bytes.

    make -C oracle ref && python tests/golden/make_stage_golden.py
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from tests import oracle_lib as ol  # noqa: E402

HARNESS = os.path.join(REPO, "oracle", "_ref", "ref_harness")
OUT = os.path.join(HERE, "stages")
THRESHOLDS = (0.3, 1.0, 2.5)
# name -> (T, levels of the trace, data seed, weight multiplier)
CASES = {
    "s13": (13, 2, 31, 1.0), "s16": (16, 2, 32, 1.0), "s1000": (1000, 3, 33, 1.0), "s4096": (4096, 3, 34, 1.0),
    "s5000_m": (5000, 3, 35, 1.5), "s65534": (65534, 3, 36, 1.0), "s65535": (65535, 3, 37, 1.0),
    "s65536": (65536, 3, 38, 1.0), "s65537": (65537, 3, 39, 1.0), "s100000": (100000, 3, 40, 1.0),
    "s131073": (131073, 4, 41, 1.0),
}
BIG = 8192


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def parse(path):
    recs = []
    with open(path, "rb") as f:
        while True:
            head = f.read(28)
            if len(head) < 28:
                break
            name = head[:16].split(b"\0")[0].decode()
            count, elem = struct.unpack("<QI", head[16:])
            recs.append((name, f.read(count * elem), elem))
    return recs


def main():
    if not os.path.exists(HARNESS):
        raise SystemExit("harness missing: run `make -C oracle ref` in the build container")
    os.makedirs(OUT, exist_ok=True)
    for name, (T, levels, seed, mult) in CASES.items():
        x = ol.trace(T, levels, seed)
        with tempfile.TemporaryDirectory() as tmp:
            fin, fout = os.path.join(tmp, "x.f32"), os.path.join(tmp, "out.bin")
            x.tofile(fin)
            subprocess.run([HARNESS, fin, fout, repr(mult)] + ["%.9g" % t for t in THRESHOLDS], check=True)
            recs = parse(fout)
        fx = {"T": np.int64(T), "levels": np.int64(levels), "seed": np.int64(seed), "mult": np.float32(mult),
              "thresholds": np.array(THRESHOLDS, np.float32)}
        k = -1
        for rname, raw, elem in recs:
            dt = {"sigma": np.float64, "starts": np.uint32}.get(rname, np.float32)
            a = np.frombuffer(raw, dt)
            if rname == "thr":
                k += 1
                continue
            key = rname if k < 0 else "%s_%d" % (rname, k)
            if a.size > BIG:
                fx[key + "_sha256"] = digest(a)
                fx[key + "_size"] = np.int64(a.size)
            else:
                fx[key] = a
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
        print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in fx.items() if k.startswith("starts")})


if __name__ == "__main__":
    main()
