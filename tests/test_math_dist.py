"""Building blocks shared by the kernels and the checker: Philox, transcendental functions, variates."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as ol


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    lib = ol.load()
    out = (C.c_uint32 * 4)()
    lib.orc_philox(0, 0, 0, 0, 0, 0, out)
    assert list(out) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    lib.orc_philox(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, out)
    assert list(out) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    lib.orc_philox(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0, out)
    assert list(out) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_expf_matches_host_libm_on_the_sampler_range():
    """hml_expf restates glibc's expf algorithm; on x <= 0 (every argument the sampler produces) a
    sample of 2^26 bit patterns matches the host libm except for isolated last-bit cases (the host
    library's FMA variant); full-range sweep: tools/check_expf_exhaustive.py."""
    lib = ol.load()
    first = C.c_uint32(0)
    # negative floats: bit patterns 0x80000000 .. 0xc2d00000 (-104); sample 64 slices of 2^20
    bad = 0
    n = 0
    for k in range(64):
        lo = 0x80000000 + k * ((0xc2d00000 - 0x80000000) // 64)
        bad += lib.orc_expf_mismatches(lo, lo + (1 << 20) - 1, C.byref(first))
        n += 1 << 20
    assert bad <= 2, bad


def test_logf_powf_within_one_ulp_of_libm():
    lib = ol.load()
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(-80, 80, 1 << 20)).astype(np.float32)
    a = np.empty_like(x)
    b = np.empty_like(x)
    lib.orc_logf_dev(x.ctypes.data, a.ctypes.data, x.size)
    lib.orc_logf_libm(x.ctypes.data, b.ctypes.data, x.size)
    d = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    u = rng.uniform(0, 1, 1 << 20).astype(np.float32)
    u[u == 0] = 0.5
    p = rng.uniform(0.5, 20, 1 << 20).astype(np.float32)
    lib.orc_powf_dev(u.ctypes.data, p.ctypes.data, a.ctypes.data, x.size)
    lib.orc_powf_libm(u.ctypes.data, p.ctypes.data, b.ctypes.data, x.size)
    d = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3


def test_restated_variates_equal_libstdcxx():
    """hml_dist.h vs std::gamma_distribution / normal_distribution / discrete_distribution on the same
    mt19937 stream: identical draws and identical engine consumption."""
    lib = ol.load()
    for seed, alpha, beta in [(1, 0.5, 1.0), (2, 2.0, 0.37), (3, 12345.5, 1.0), (4, 1.0, 2.0), (5, 0.01, 1.0), (6, 0.999, 3.0)]:
        assert lib.orc_check_gamma(seed, 100000, alpha, beta) == 0
    assert lib.orc_check_normal(7, 100000, 0.3, 2.0) == 0
    for K in (2, 3, 5, 10, 16):
        # every 17th row is all zeros: libstdc++ then returns index 0 (NaN cumulative probabilities)
        assert lib.orc_check_categorical(11, 100000, K, 17) == 0


def test_synthetic_trace_is_deterministic_and_plausible():
    x1 = ol.synth_gauss(200000, 5, ol.LEVELS[5], 0.3, 5000.0, 9, nthreads=1)
    x2 = ol.synth_gauss(200000, 5, ol.LEVELS[5], 0.3, 5000.0, 9, nthreads=7)
    assert np.array_equal(x1.view(np.uint32), x2.view(np.uint32))
    lib = ol.load()
    st = np.empty(200000, np.int16)
    mu = np.asarray(ol.LEVELS[5], np.float32)
    x3 = np.empty(200000, np.float32)
    lib.orc_synth_gauss(x3.ctypes.data, st.ctypes.data, 200000, 5, mu.ctypes.data, 0.3, 5000.0, 9, 3)
    assert np.array_equal(x1.view(np.uint32), x3.view(np.uint32))
    resid = x1 - mu[st]
    assert 0.29 < resid.std() < 0.31 and abs(resid.mean()) < 0.01
    jumps = np.count_nonzero(np.diff(st))
    assert 15 < jumps < 80          # mean dwell 5000 -> about 40 level changes
    assert np.all(st[1:][np.diff(st) != 0] != st[:-1][np.diff(st) != 0])


def test_depth_trace_is_deterministic_and_plausible():
    x1, st = ol.synth_depth(400000, seed=5, nthreads=1, with_states=True)
    x2 = ol.synth_depth(400000, seed=5, nthreads=6)
    assert np.array_equal(x1.view(np.uint32), x2.view(np.uint32))
    assert np.all(x1 >= 0) and np.all(x1 == np.round(x1))
    dip = x1[st == 2]
    assert 29.0 < dip.mean() < 31.0            # depth 15 x copy number 2
    assert 1.0 < dip.var() / dip.mean() < 2.5  # over-dispersed relative to Poisson
    assert set(np.unique(st)).issubset({0, 1, 2, 3, 4})


@pytest.mark.parametrize("T,K,dseed,seed", [(100000, 5, 9, 3), (60000, 3, 1, 1), (50000, 10, 4, 7), (30000, 16, 2, 11)])
def test_emission_terms_of_device_math_within_1e6_of_reference_math(T, K, dseed, seed):
    """BASELINE.json's tolerance: emission log-likelihoods within 1e-6 relative.  E_s of one sweep computed with the
    device's arithmetic (hml_math.h) against glibc's, on the same parameters and therefore the same block structure."""
    x = ol.trace(T, K, dseed)
    res = []
    params = None
    for math, red in ((ol.MATH_DEV, ol.REDUCE_DEV), (ol.MATH_LIBM, ol.REDUCE_REF)):
        o = ol.OracleChain(K=K, seed=seed, rng=ol.RNG_CTR, math=math, reduce=red)
        o.load(x)
        o.autoprior()
        o.init_model()
        o.token("F")
        if params is None:
            params = (o.theta(), *o.transitions())
        else:
            o.set_params(*params)
        o.set_probes(True)
        o.iterate("F", 1, 0)
        res.append((o.blocks(), o.loglik()))
    assert np.array_equal(res[0][0], res[1][0])
    Ed, Er = res[0][1], res[1][1]
    assert Ed.shape == Er.shape and Ed.size >= K
    rel = np.abs(Er.astype(np.float64) - Ed) / np.maximum(np.abs(Er), 1e-30)
    assert rel.max() <= 1e-6, rel.max()
