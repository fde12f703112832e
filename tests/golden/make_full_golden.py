#!/usr/bin/env python3
"""Full-size golden files: the UNMODIFIED reference binary (oracle/_ref/hammlet, built by oracle/Makefile from
/root/reference/src/main.cpp) on the traces of BASELINE.json's configs 2, 3 and 4 - 10^7 and 10^8 positions - so that
the regimes that exist only at scale are held by reference-written files and not by the checker alone: per-state
counts above 2^24 (`size_t += float` rounds there, src/StateSequence/ForwardBackward.hpp:183-187), ~1526 cells of the
integral array (src/Statistics/IntegralArray.hpp:136-191), blocks beyond the uint16 pointer range
(src/Blocks/BreakpointArray.hpp:130-184), the recording of 10^5-block sweeps (src/StateMarginals.hpp:268-310).

Only runs where /root/reference exists (the build container).  Committed per case under tests/golden/full/<case>/:
the reference's `marginals`, `parameters` and `compression` files (xz-compressed where larger than 1 MB - data, not
source; a file above 256 MB would be kept as its sha256, size and line count only - config 4's chain draws a state
of variance 0 in its first sweeps, the threshold falls to 0, every position becomes a block, and the marginals are
4.7 10^6 segments: 104 MB, 3 MB compressed), the sha256 of the float32 trace the tests regenerate from (T, levels, data seed) with the
repository's generator, and the sha256 of every uncompressed output.  Configs 2 and 3 take two minutes; config 4
takes 70 minutes in the reference binary (80 s per sweep of 10^8 blocks) and 6 GB.

    python tests/golden/make_full_golden.py [case ...]
    python tests/golden/make_full_golden.py --from-dir DIR case      (outputs ref-<type>.csv of a run made by hand)
"""
import ctypes as C
import hashlib
import json
import lzma
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from tests import oracle_lib as ol  # noqa: E402

REF = os.path.join(REPO, "oracle", "_ref", "hammlet")
OUT = os.path.join(HERE, "full")
OUTPUTS = ["marginals", "parameters", "compression"]

# name -> (T, levels of the trace, data seed, flags): the traces are bench.py's WORKLOADS for the same configs
CASES = {
    # config 2: mixture burn-in, then a FIXED block structure (S), a fresh prior draw and recorded FB sweeps
    "c2_1e7_k5_static": (10_000_000, 5, 2, "-s 5 -R 1 -i M 100 0 S P F 200 10"),
    # config 3: dynamic recompression every sweep, every 10th recorded
    "c3_1e8_k5_dynamic": (100_000_000, 5, 3, "-s 5 -R 1 -i F 50 10"),
    # config 4's trace and model (one of its eight chains)
    "c4_1e8_k10_dynamic": (100_000_000, 10, 4, "-s 10 -R 1 -i F 50 10"),
}


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for piece in iter(lambda: f.read(1 << 24), b""):
            h.update(piece)
    return h.hexdigest()


def keep_outputs(name, srcdir, x, dt, manifest):
    """the reference's output files ref-<type>.csv under srcdir -> tests/golden/full/<name>/ and the manifest entry"""
    T, K, dseed, flags = CASES[name]
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    entry = {"T": T, "trace_levels": K, "data_seed": dseed, "flags": flags, "outputs": OUTPUTS,
             "trace_sha256": hashlib.sha256(x.tobytes()).hexdigest(), "reference_seconds": round(dt, 1), "files": {}}
    for o in OUTPUTS:
        src = os.path.join(srcdir, "ref-%s.csv" % o)
        size = os.path.getsize(src)
        for old in (o + ".csv", o + ".csv.xz"):
            if os.path.exists(os.path.join(d, old)):
                os.remove(os.path.join(d, old))
        h, lines, head = hashlib.sha256(), 0, b""
        with open(src, "rb") as f:
            for piece in iter(lambda: f.read(1 << 24), b""):
                h.update(piece)
                lines += piece.count(b"\n")
                if len(head) < 4096:
                    head += piece[:4096 - len(head)]
        fn = None
        if size <= (1 << 20):
            fn = o + ".csv"
            with open(src, "rb") as f, open(os.path.join(d, fn), "wb") as g:
                g.write(f.read())
        elif size <= (256 << 20):
            fn = o + ".csv.xz"
            with open(src, "rb") as f, open(os.path.join(d, fn), "wb") as g:
                g.write(lzma.compress(f.read(), preset=9 | lzma.PRESET_EXTREME))
        entry["files"][o] = {"file": fn, "bytes": size, "sha256": h.hexdigest(), "lines": lines}
        if fn is None:   # too large to keep: the checksum, and the first lines for a reader
            entry["files"][o]["first_lines"] = head.decode().split("\n")[:8]
    manifest[name] = entry
    print(name, "ok: %.1f s in the reference binary" % dt, {o: entry["files"][o]["bytes"] for o in OUTPUTS}, flush=True)


def main():
    if not os.path.exists(REF):
        raise SystemExit("reference binary missing: run `make -C oracle ref` in the build container")
    lib = ol.load()
    lib.orc_write_text.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_int]
    mpath = os.path.join(OUT, "manifest.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 3 and sys.argv[1] == "--from-dir":
        name = sys.argv[3]
        T, K, dseed, flags = CASES[name]
        keep_outputs(name, sys.argv[2], ol.trace(T, K, dseed), float(os.environ.get("HML_REFERENCE_SECONDS", "0")), manifest)
        with open(mpath, "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    names = sys.argv[1:] or list(CASES)
    for name in names:
        T, K, dseed, flags = CASES[name]
        x = ol.trace(T, K, dseed)
        with tempfile.TemporaryDirectory(dir=os.environ.get("HML_GOLDEN_TMP") or None) as tmp:
            inp = os.path.join(tmp, "in.txt")
            assert lib.orc_write_text(x.ctypes.data, x.size, inp.encode(), 8) == 0
            t0 = time.perf_counter()
            cmd = [REF, "-f", inp, "-o", os.path.join(tmp, "ref-"), ".csv", "-w", "-a"] + flags.split() + ["-O"] + OUTPUTS
            r = subprocess.run(cmd, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            keep_outputs(name, tmp, x, time.perf_counter() - t0, manifest)
        with open(mpath, "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
