"""hml_math_glibc.h (the reference-compatible mode's expf / logf / powf: glibc 2.35's algorithms in its FMA build) against
this host's libm, which is what the reference binary calls.  The whole ranges take two minutes (run once per change:
HML_FULL_MATH_SWEEP=1; results in DESIGN.md); the default run takes slices around every structural edge."""
import ctypes as C
import os

import pytest

from tests import oracle_lib as ol


def mismatches(fn, lo, hi):
    lib = ol.load()
    lib.orc_glibc_mismatches.restype = C.c_int64
    lib.orc_glibc_mismatches.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
    first = (C.c_uint32 * 2)()
    n = lib.orc_glibc_mismatches(fn, lo, hi, first)
    if n < 0:
        pytest.skip("this CPU has no fused multiply-add: its libm runs glibc's other variant")
    return n, (hex(first[0]), hex(first[1]))


FULL = os.environ.get("HML_FULL_MATH_SWEEP") == "1"


def test_expf_is_glibcs():
    if FULL:
        assert mismatches(0, 0, 1 << 32)[0] == 0
        return
    # the two inputs on which the separately rounded form (hml_expf) differs, the special-case edges, sub-normal results
    for lo, hi in ((0x4202422f - 1000, 0x4202422f + 1000), (0xc27c65d9 - 1000, 0xc27c65d9 + 1000), (0x42b00000, 0x42b20000),
                   (0xc2cf0000, 0xc2d10000), (0xc2ae0000, 0xc2b00000), (0x7f7fff00, 0x7f800100), (0xff7fff00, 0xff800100),
                   (0, 1 << 20), (0x80000000, 0x80000000 + (1 << 20)), (0x3f000000, 0x3f000000 + (1 << 22)), (0xbf000000, 0xbf000000 + (1 << 22))):
        n, first = mismatches(0, lo, hi)
        assert n == 0, (hex(lo), first)


def test_logf_is_glibcs():
    if FULL:
        assert mismatches(1, 0, (1 << 31) + (1 << 20))[0] == 0
        return
    for lo, hi in ((0, 1 << 20), (0x00800000 - 4096, 0x00800000 + 4096), (0x3f330000 - (1 << 18), 0x3f330000 + (1 << 18)),
                   (0x3f800000 - (1 << 20), 0x3f800000 + (1 << 20)), (0x7f7ff000, 0x7f800001), (0x2f000000, 0x2f000000 + (1 << 21)),
                   (0x4b000000, 0x4b000000 + (1 << 21))):
        n, first = mismatches(1, lo, hi)
        assert n == 0, (hex(lo), first)


def test_powf_is_glibcs_where_the_sampler_uses_it():
    """x in [0, 1] (a canonical uniform or any pattern), y > 0 (1 / alpha of a gamma_distribution with alpha < 1, or any)"""
    n, first = mismatches(2, 11, (1 << 31) if FULL else (1 << 23))
    assert n == 0, first
