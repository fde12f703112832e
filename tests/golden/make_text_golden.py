#!/usr/bin/env python3
"""Golden vectors for the text reader from the UNMODIFIED reference reader: writes small text files and lets
oracle/_ref/ref_harness --parse (the reference's MaxletTransform, src/wavelet.hpp:97-134, compiled from the
reference's headers by oracle/Makefile) extract their values.  Commits tests/golden/text/NAME.txt (input) and
NAME.f32 (the float32 values the reference extracted, in order).  Only runs where /root/reference exists.

    make -C oracle ref && python tests/golden/make_text_golden.py
"""
import os
import random
import struct
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(REPO, "oracle", "_ref", "ref_harness")
OUT = os.path.join(HERE, "text")


def f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def cases():
    rnd = random.Random(20240607)
    c = {}
    # plain measurement-like columns in the formats tools write
    vals = [rnd.gauss(0, 1) * 10 ** rnd.randint(-3, 4) for _ in range(3000)]
    fmts = ["%.3f", "%.6g", "%.9g", "%.17g", "%e", "%.1f", "%+.4f", "%.12E", "%d"]
    lines = []
    for v in vals:
        f = rnd.choice(fmts)
        lines.append(f % (int(v) if f == "%d" else v))
    c["formats"] = "\n".join(lines) + "\n"
    c["separators"] = "1 2\t3\n4\r\n5\v6\f7   8\n\n\n9 \t\r\n10"
    c["signs_dots"] = "+1 -1 +.5 -.5 5. -5. +5.e1 .5e+1 0.0 -0 +0 -0.0 00012 -000.5 0e5 0e-999 1e0 1E2 1e+2 1e-2 0001e0002\n"
    # ties, near-ties and long digit strings (decided by the host extraction in the build)
    hard = []
    for _ in range(300):
        b = rnd.randrange(0x00800000, 0x7f000000)
        lo, hi = f32(b), f32(b + 1)
        mid = (lo + hi) / 2  # exact in double
        hard.append("%.40g" % mid)
        hard.append("%.19g" % lo)
        hard.append("%.20g" % hi)
        hard.append("%.9g" % lo)
    hard += ["16777217", "16777219", "33554434", "47378058", "8388608.5", "8388609.5", "9007199254740993", "18446744073709551615",
             "18446744073709551616", "123456789012345678901234567890", "0." + "0" * 60 + "1", "1" + "0" * 38, "3.4028235e38",
             "3.4028234663852886e38", "3.4028235677973366e38", "1.17549435e-38", "1.1754942e-38", "1e-45", "1.4e-45", "7e-46",
             "6e-46", "1e-60", "-1e-60", "0.1", "0.2", "0.3", "1e22", "1e23", "1e-22", "1e-23", "9999999999999999999",
             "4.35", "4.349999999999999", "2.5e-1", "0.5000000298023223876953125", "0.50000002980232238769531250001"]
    rnd.shuffle(hard)
    c["hard_values"] = " ".join(hard) + "\n"
    # tokens that are several values, or a value followed by a failure
    c["glued"] = "1.5-3 2+4 1.2.3 7-8-9\n10 11"
    c["stop_word"] = "1 2 3 abc 4 5\n"
    c["stop_nan"] = "1 2 nan 3\n"
    c["stop_inf"] = "0.5 inf 3\n"
    c["stop_hex"] = "7 0x10 3\n"
    c["stop_overflow"] = "1 2 1e39 3\n"
    c["stop_neg_overflow"] = "1 -3.5e38 3\n"
    c["stop_bare_exp"] = "1 5e 3\n"
    c["stop_exp_sign"] = "1 5e+ 3\n"
    c["stop_dot"] = "1 . 3\n"
    c["stop_sign"] = "1 - 3\n"
    c["stop_comma"] = "1,5 2\n"
    c["stop_tail_garbage"] = "1 2 3x"
    c["stop_last_bare_exp"] = "1 2 5e"
    c["stop_first"] = "x 1 2\n"
    c["header_line"] = "value\n1\n2\n"
    c["empty"] = ""
    c["blanks_only"] = " \n\t \n"
    c["one"] = "42"
    c["long_token"] = "1 " + "0" * 100 + "1." + "5" * 80 + " 2 " + "9" * 70 + "e-60 3\n"
    c["exp_huge"] = "1 1e99999999999999999999 2\n"
    c["exp_huge_zero"] = "1 0e99999999999999999999 2\n"
    c["exp_tiny"] = "1 1e-99999999999999999999 2\n"
    # a few hundred KB so that chunked reading with small staging buffers crosses many boundaries
    big = []
    for i in range(60000):
        v = rnd.gauss(2, 0.7)
        big.append(("%.4f" % v) if i % 3 else ("%.7g" % (v * 1e3)))
    c["column_60k"] = "\n".join(big) + "\n"
    return c


def main():
    if not os.path.exists(HARNESS):
        raise SystemExit("harness missing: run `make -C oracle ref` in the build container")
    os.makedirs(OUT, exist_ok=True)
    for name, text in cases().items():
        txt = os.path.join(OUT, name + ".txt")
        with open(txt, "w", newline="") as f:
            f.write(text)
        out = os.path.join(OUT, name + ".f32")
        subprocess.run([HARNESS, "--parse", txt, out], check=True)
        print("%-22s %8d bytes -> %6d values" % (name, len(text), os.path.getsize(out) // 4))


if __name__ == "__main__":
    main()
