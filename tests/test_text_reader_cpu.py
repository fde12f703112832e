"""Text reader, CPU side: the checker's restatement of the reference's reader (`while ( input >> v )`, reference
src/wavelet.hpp:131) against the golden values the reference's own reader extracted (tests/golden/text, made by
tests/golden/make_text_golden.py), and the product's token converter (hml_text.h compiled by gcc) against strtof."""
import glob
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

TEXT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(TEXT_DIR, "*.txt")))


def golden(name):
    with open(os.path.join(TEXT_DIR, name + ".txt"), "rb") as f:
        text = f.read()
    return text, np.fromfile(os.path.join(TEXT_DIR, name + ".f32"), np.float32)


def stops_early(name):
    """cases in which an extraction fails before the end of the text (the reference stops reading there)"""
    return name.startswith("stop_") or name in ("header_line", "exp_huge")


def test_there_are_goldens():
    assert len(CASES) >= 25


@pytest.mark.parametrize("name", CASES)
def test_checker_reader_matches_the_reference_reader(name):
    text, want = golden(name)
    got, stopped = ol.parse_text(text)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert stopped == stops_early(name)


def random_tokens(rng, n):
    toks = []
    for _ in range(n):
        kind = rng.integers(0, 7)
        if kind == 0:
            f = np.array([rng.integers(0, 2 ** 32)], np.uint32).view(np.float32)[0]
            if not np.isfinite(f):
                f = np.float32(1.5)
            toks.append("%.*g" % (int(rng.integers(1, 20)), float(f)))
        elif kind == 1:
            toks.append("%.*f" % (int(rng.integers(0, 10)), rng.normal() * 10.0 ** rng.integers(-3, 5)))
        elif kind == 2:
            toks.append("%d" % (int(rng.integers(0, 2 ** 63)) >> int(rng.integers(0, 63))))
        elif kind == 3:
            toks.append("%de%d" % (int(rng.integers(0, 2 ** 63)) >> int(rng.integers(0, 63)), int(rng.integers(-70, 50))))
        elif kind == 4:   # midpoints of neighbouring floats, exact decimal expansion cut at 1..25 digits
            b = int(rng.integers(0x00800000, 0x7f000000))
            lo, hi = np.array([b, b + 1], np.uint32).view(np.float32)
            toks.append("%.*g" % (int(rng.integers(1, 26)), (float(lo) + float(hi)) / 2))
        elif kind == 5:
            toks.append("%s.%dE%+d" % ("+-"[int(rng.integers(0, 2))], int(rng.integers(0, 10 ** 9)), int(rng.integers(-40, 40))))
        else:
            toks.append(["1.", ".5", "-.25e1", "+0", "-0.0", "000123", "1e+05", "5e", ".", "-", "abc", "1,5", "0x1p3", "nan", "1.5-3",
                         "16777217", "8388608.5", "1e39", "1e-46", "123456789012345678901"][int(rng.integers(0, 20))])
    return toks


def test_token_converter_agrees_with_strtof_whenever_it_decides():
    rng = np.random.default_rng(7)
    toks = random_tokens(rng, 400000)
    out, status, ref = ol.parse_tokens(toks)
    decided = status == 0
    assert decided.mean() > 0.6
    bad = np.nonzero(decided & (out.view(np.uint32) != ref.view(np.uint32)))[0]
    assert bad.size == 0, [(toks[i], out[i], ref[i]) for i in bad[:5]]
    # tokens that are not one plain decimal number are never decided by the converter
    for t, s in zip(toks, status):
        if t in ("5e", ".", "-", "abc", "1,5", "0x1p3", "nan", "1.5-3", "1e39", "1e-46", "123456789012345678901"):
            assert s == 1, t
    # ... and the everyday formats always are
    out, status, ref = ol.parse_tokens(["%.4f" % v for v in rng.normal(size=20000)] + ["%.9g" % v for v in rng.normal(size=20000)])
    assert status.sum() == 0
