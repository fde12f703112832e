#!/usr/bin/env python3
"""Golden outputs of the reference's bin/sortStates script (run with LC_ALL=C from /root/reference/bin) for every committed
golden parameters file: tests/golden/*/sortstates.txt.  Only runs where /root/reference exists.

    python tests/golden/make_sortstates_golden.py
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = "/root/reference/bin/sortStates"


def main():
    if not os.path.exists(SCRIPT):
        raise SystemExit("reference script missing")
    for path in sorted(glob.glob(os.path.join(HERE, "*", "parameters.csv"))):
        out = os.path.join(os.path.dirname(path), "sortstates.txt")
        with open(out, "w") as f:
            subprocess.run(["bash", SCRIPT, path], check=True, stdout=f, env=dict(os.environ, LC_ALL="C"))
        print(os.path.relpath(out, HERE), sum(1 for _ in open(out)), "lines")


if __name__ == "__main__":
    main()
