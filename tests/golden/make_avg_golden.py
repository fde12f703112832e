#!/usr/bin/env python3
"""Golden outputs of the reference's window-average tool: runs oracle/_ref/avg (built by oracle/Makefile from
/root/reference/src/tools/avg.cpp) on committed text inputs (tests/golden/text/*.txt) with several window sizes and
commits its standard output as tests/golden/avg/NAME.wWINDOW.out.  Only runs where /root/reference exists.

    make -C oracle ref && python tests/golden/make_avg_golden.py
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
TOOL = os.path.join(REPO, "oracle", "_ref", "avg")
CASES = [("column_60k", 1000), ("column_60k", 7), ("column_60k", 100000), ("formats", 50), ("separators", 3), ("stop_word", 2),
         ("glued", 4), ("empty", 5), ("hard_values", 100), ("one", 1), ("column_60k", 0)]


def main():
    if not os.path.exists(TOOL):
        raise SystemExit("reference tool missing: run `make -C oracle ref` in the build container")
    out_dir = os.path.join(HERE, "avg")
    os.makedirs(out_dir, exist_ok=True)
    for name, window in CASES:
        with open(os.path.join(HERE, "text", name + ".txt"), "rb") as f, open(os.path.join(out_dir, "%s.w%d.out" % (name, window)), "wb") as g:
            subprocess.run([TOOL, str(window)], stdin=f, stdout=g, check=True)
        print(name, window, "ok")


if __name__ == "__main__":
    main()
