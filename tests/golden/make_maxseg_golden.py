#!/usr/bin/env python3
"""Golden outputs of the reference's post-processing tool: runs oracle/_ref/maxSegmentation (built by oracle/Makefile
from /root/reference/src/tools/maxSegmentation.cpp) on every committed golden marginals file and on a few hand-made
edge cases, and commits the tool's standard output next to the input.  Only runs where /root/reference exists.

    make -C oracle ref && python tests/golden/make_maxseg_golden.py
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
TOOL = os.path.join(REPO, "oracle", "_ref", "maxSegmentation")

EDGE = {
    "first_state_nonzero": "5\t0\t3\n7\t0\t9\n2\t4\t1\n",
    "all_zero_rows": "5\t0\t0\n7\t0\t0\n",
    "ties_first_maximum": "3\t2\t2\t1\n4\t1\t5\t5\n",
    "single_line": "10\t1\t2\t3\n",
    "empty": "",
    "no_counts": "100\n",
    "merge_everything": "1\t3\t0\n2\t4\t1\n3\t9\t8\n",
    "alternating": "".join("%d\t%d\t%d\n" % (i + 1, i % 2, (i + 1) % 2) for i in range(12)),
}


def main():
    if not os.path.exists(TOOL):
        raise SystemExit("reference tool missing: run `make -C oracle ref` in the build container")
    edge_dir = os.path.join(HERE, "maxseg")
    os.makedirs(edge_dir, exist_ok=True)
    for name, text in EDGE.items():
        with open(os.path.join(edge_dir, name + ".marginals"), "w") as f:
            f.write(text)
    inputs = sorted(glob.glob(os.path.join(HERE, "*", "marginals.csv"))) + sorted(glob.glob(os.path.join(edge_dir, "*.marginals")))
    for path in inputs:
        out = path[:-len("marginals.csv")] + "maxsegmentation.txt" if path.endswith("marginals.csv") else path[:-len(".marginals")] + ".maxseg"
        with open(out, "w") as f:
            subprocess.run([TOOL, "-i", path], check=True, stdout=f)
        print(os.path.relpath(out, HERE), sum(1 for _ in open(out)), "lines")


if __name__ == "__main__":
    main()
