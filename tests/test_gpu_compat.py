"""The reference-compatible mode on the GPU (hml_set_option "compat", `hammlet -compat`; hml_k_compat.h): with the
reference's seed the GPU leaves the REFERENCE'S files - BASELINE.json's "state-marginal counts at a fixed RNG seed" met
literally, against golden files written by the unmodified reference binary (tests/golden/, make_golden.py), not through
the checker's device mode.  Also: its arithmetic on the device against the host's libm."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "hammlet_amd", "hammlet")
GOLD = os.path.join(REPO, "tests", "golden")
with open(os.path.join(GOLD, "manifest.json")) as f:
    MANIFEST = json.load(f)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def libm(fn, a, b=None):
    lib = ol.load()
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    if b is None:
        getattr(lib, fn)(a.ctypes.data, out.ctypes.data, a.size)
    else:
        b = np.ascontiguousarray(b, np.float32)
        getattr(lib, fn)(a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
    return out


def test_glibc_arithmetic_on_the_device(hml):
    """hml_glibc_expf / logf / powf_unit evaluated on the GPU against std::exp / std::log / std::pow of this host (glibc;
    the container's whole-range comparison is tests/test_math_glibc_cpu.py)"""
    rng = np.random.default_rng(5)
    n = 1 << 22
    x = np.concatenate([rng.normal(0, 30, n), -rng.exponential(20, n), [0.0, -0.0, 88.72, 88.73, -103.9, -104.0, -87.3, np.inf, -np.inf],
                        np.array([0x4202422f, 0xc27c65d9], np.uint32).view(np.float32)]).astype(np.float32)
    assert np.array_equal(bits(hml.debug_eval(40, x)), bits(libm("orc_expf_libm", x)))
    p = np.concatenate([np.exp(rng.uniform(-100, 88, n)), rng.uniform(0.5, 2.0, n), [1.0, 0.0, np.inf, 1e-45, 1.17549435e-38, 0.7]]).astype(np.float32)
    assert np.array_equal(bits(hml.debug_eval(41, p)), bits(libm("orc_logf_libm", p)))
    u = np.concatenate([rng.integers(0, 1 << 24, n).astype(np.float32) / np.float32(1 << 24), [0.0, 1.0, 1e-45, 1e-39]]).astype(np.float32)
    y = (1.0 / np.concatenate([rng.uniform(0.001, 1.0, n), [0.5, 0.5, 0.25, 0.9]])).astype(np.float32)
    assert np.array_equal(bits(hml.debug_eval(42, u, y)), bits(libm("orc_powf_libm", u, y)))


def test_quotient_by_reciprocal(hml):
    """hml_tr2_quotient (hml_k_trellis_rows.h): f / Z through one double reciprocal and a correction step equals the
    division itself - random pairs, sub-normal quotients and constructed exact ties on the sub-normal grid
    (tools/div_check.hip runs 3.4 10^10 pairs)"""
    rng = np.random.default_rng(9)
    n = 1 << 22
    Z = rng.integers(1, 0x7f7fffff, n, dtype=np.uint32).view(np.float32)
    f = (rng.integers(0, 0x7f7fffff, n, dtype=np.uint32).view(np.float32))
    f = np.minimum(f, Z)
    b = 2 * rng.integers(0, 2048, n) + 1
    n2 = rng.integers(0, 2048, n)
    ok = b * (2 * n2 + 1) < (1 << 24)
    e = rng.integers(-20, 20, n)
    Zt = np.ldexp(b.astype(np.float64), e).astype(np.float32)[ok]
    ft = np.ldexp((b * (2 * n2 + 1)).astype(np.float64), e - 150).astype(np.float32)[ok]
    Zs = np.exp(rng.uniform(-40, 40, n)).astype(np.float32)
    fs = (Zs.astype(np.float64) * np.ldexp(1.0 + rng.random(n), -rng.integers(126, 151, n))).astype(np.float32)
    for ff, zz in ((f, Z), (ft, Zt), (fs, Zs)):
        assert np.array_equal(bits(hml.debug_eval(43, ff, zz)), bits(hml.debug_eval(4, ff, zz)))


@pytest.mark.parametrize("T,K,seed,scheme", [
    (100000, 3, 1, [("F", 30, 1)]),
    (20000, 4, 3, [("M", 20, 5), ("D",), ("F", 25, 2), ("P",), ("M", 6, 1), ("S",), ("F", 12, 1)]),
    (70000, 5, 8, [("F", 12, 3)]),
    # more than 16 states (round 4): the mode takes the number of states at run time, up to 64
    (50000, 17, 9, [("F", 10, 1)]),
    (30000, 33, 10, [("M", 6, 2), ("S",), ("F", 8, 1), ("P",), ("D",), ("F", 5, 1)]),
    (20000, 64, 11, [("F", 6, 1)]),
])
@pytest.mark.parametrize("chunks,warmup", [(None, None), (37, -1), (200, 3), (1, None), ("cap64", -1)])
def test_compat_chain_is_the_reference_chain(hml, monkeypatch, T, K, seed, scheme, chunks, warmup):
    """through the C ABI: block structure, state sequence, parameter bits, transition matrix, counts and marginals of a
    compat chain equal those of the checker in REFERENCE mode (sequential mt19937, libm, Kahan sums, size_t += float).
    Round 4: filter and backward draws run in chunks that are checked against each other, the count pass by state
    (hml_k_compat.h): the default geometry; 37 chunks without any warm-up - every chunk starts from a flat row / state 0, so
    chunks ARE wrong and run again (the statistic says so) - and 200 with 3 blocks of it; one chunk, the sequential form."""
    if chunks == "cap64":   # ... and per-block buffers for 64 blocks: the chain halts, the host grows them and sweeps again (hml_settle) -
        monkeypatch.setenv("HML_MAX_BLOCKS", "64")   # the engine's outputs of a halted sweep must not be consumed
        chunks = 29
    if chunks is not None:
        monkeypatch.setenv("HML_COMPAT_CHUNKS", str(chunks))
    if warmup is not None:
        monkeypatch.setenv("HML_COMPAT_WARMUP", str(warmup))
    x = ol.trace(T, min(K, 6), seed)
    o = ol.OracleChain(K=K, seed=seed, rng=ol.RNG_MT, math=ol.MATH_LIBM, reduce=ol.REDUCE_REF)
    o.load(x)
    o.autoprior()
    o.init_model()
    o.set_record(marginals=True)
    g = hml.Chain(device=0, seed=seed)
    g.set_option("compat", 1)
    g.load(x)
    g.set_model(K, g.autoprior(0.2, 0.9))
    pending = True
    for tok in scheme:
        o.token(tok[0])
        if pending:
            g.sample_prior()
            pending = False
        if tok[0] == "P":
            pending = True
        elif tok[0] == "S":
            g.set_static_blocks()
        elif tok[0] == "D":
            g.set_dynamic(True)
        else:
            o.iterate(tok[0], tok[1], tok[2])
            g.iterate(tok[0], tok[1], tok[2])
            g.sync()
            assert np.array_equal(o.blocks(), g.blocks())
            assert np.array_equal(o.states(), g.states())
            assert np.array_equal(bits(o.theta()), bits(g.theta()))
            Ao, pio = o.transitions()
            Ag, pig = g.transitions()
            assert np.array_equal(bits(Ao), bits(Ag)) and np.array_equal(bits(pio), bits(pig))
    seg, cnt = g.marginals_rle()
    assert hml.marginals_text(seg, cnt) == o.text("marginals")
    if warmup == -1:
        assert g.stats()["forward_refits"] > 0   # chunks were wrong and ran again


@pytest.mark.parametrize("P,D,T,seed,scheme", [
    (6, 2, 20000, 8, [("F", 8, 1)]),   # 36 states
    (2, 2, 30000, 5, [("F", 20, 1)]),
    (3, 2, 20000, 6, [("M", 10, 1), ("S",), ("P",), ("F", 12, 2), ("D",), ("F", 6, 1)]),
    (2, 3, 20000, 7, [("F", 10, 1)]),
])
def test_compat_multivariate_chain_is_the_reference_chain(hml, P, D, T, seed, scheme):
    """`-s C P D` in the reference-compatible mode through the C ABI: per-parameter Kahan sums in dimension order, the
    state's log-normaliser as the float sum of its parameters', theta drawn per parameter from the one mt19937 stream -
    blocks, states, parameter bits, transition matrix and marginals equal the checker's REFERENCE mode."""
    K = P ** D
    x = np.stack([ol.trace(T, P, seed + 40 + d) for d in range(D)], axis=1).reshape(-1)
    o = ol.OracleChain(K=K, seed=seed, rng=ol.RNG_MT, math=ol.MATH_LIBM, reduce=ol.REDUCE_REF)
    o.set_dimensions(D, P)
    o.load(x)
    o.autoprior()
    o.init_model()
    o.set_record(marginals=True)
    g = hml.Chain(device=0, seed=seed)
    g.set_option("compat", 1)
    g.set_dimensions(D, P)
    g.load(x)
    g.set_model(K, g.autoprior(0.2, 0.9))
    pending = True
    for tok in scheme:
        o.token(tok[0])
        if pending:
            g.sample_prior()
            pending = False
        if tok[0] == "P":
            pending = True
        elif tok[0] == "S":
            g.set_static_blocks()
        elif tok[0] == "D":
            g.set_dynamic(True)
        else:
            o.iterate(tok[0], tok[1], tok[2])
            g.iterate(tok[0], tok[1], tok[2])
            g.sync()
            assert np.array_equal(o.blocks(), g.blocks())
            assert np.array_equal(o.states(), g.states())
            assert np.array_equal(bits(o.theta()), bits(g.theta()))
            Ao, pio = o.transitions()
            Ag, pig = g.transitions()
            assert np.array_equal(bits(Ao), bits(Ag)) and np.array_equal(bits(pio), bits(pig))
    seg, cnt = g.marginals_rle()
    assert hml.marginals_text(seg, cnt) == o.text("marginals")


@pytest.mark.parametrize("chunks", [None, "50:-1"])
@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_cli_compat_writes_the_reference_binarys_files(case, chunks, monkeypatch):
    """`hammlet -compat` with the flags of a golden run against the files the UNMODIFIED REFERENCE BINARY wrote for them
    (tests/golden/<case>/, oracle/_ref/hammlet in the build container): byte for byte, from the GPU - every run of the manifest: the
    univariate ones, the four multivariate / shared-parameter runs `-s C P D` (reference src/Mapping.hpp:53-137,
    src/EFD.hpp:83-93, src/StateSequence/ForwardBackward.hpp:189-207), the runs with 20-64 states, and (round 5) the `segments`
    side file (src/Records.hpp:208-209, src/StateMarginals.hpp:204) of five of them."""
    if chunks is not None:   # every sweep in 50 chunks without warm-up: chunks that start wrong run again
        monkeypatch.setenv("HML_COMPAT_CHUNKS", chunks.split(":")[0])
        monkeypatch.setenv("HML_COMPAT_WARMUP", chunks.split(":")[1])
    m = MANIFEST[case]
    x = ol.trace(m["T"], m["trace_levels"], m["data_seed"])
    if m.get("dims", 1) > 1:   # dimension d = the generator with data seed + d, interleaved by position (tests/test_oracle_golden.py)
        x = np.stack([ol.trace(m["T"], m["trace_levels"], m["data_seed"] + d) for d in range(m["dims"])], axis=1).reshape(-1)
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "in.f32")
        x.tofile(raw)
        r = subprocess.run([CLI, "-compat", "-raw", raw, "-o", os.path.join(tmp, "g-"), ".csv", "-a"] + m["flags"].split() + ["-O"] + m["outputs"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for o in m["outputs"]:
            got = open(os.path.join(tmp, "g-%s.csv" % o)).read()
            want = open(os.path.join(GOLD, case, o + ".csv")).read()
            assert got == want, (case, o)


def test_compat_counts_above_2_24_round_like_the_reference(hml):
    """where the reference's `size_t += float` counts (ForwardBackward.hpp:183-187) exceed 2^24 they
    ROUND - the default path counts exactly (D4), so only the reference-compatible mode can be compared with the reference
    there.  4 10^7 positions, two states: occupancies of 2 10^7 each; the compat chain's counts, Kahan sums, parameters and
    states equal the checker's reference mode bit for bit, and the counts do differ from the exact ones."""
    T, K, seed = 40_000_000, 2, 4
    x = ol.synth_gauss(T, K, ol.LEVELS[K], ol.SIGMA[K], ol.DWELL[K], seed, nthreads=16)
    o = ol.OracleChain(K=K, seed=seed, rng=ol.RNG_MT, math=ol.MATH_LIBM, reduce=ol.REDUCE_REF)
    o.load(x)
    o.autoprior()
    o.init_model()
    o.token("F")
    g = hml.Chain(device=0, seed=seed)
    g.set_option("compat", 1)
    g.load(x)
    g.set_model(K, g.autoprior(0.2, 0.9))
    g.sample_prior()
    o.iterate("F", 4, 0)
    g.iterate("F", 4, 0)
    g.sync()
    assert np.array_equal(o.blocks(), g.blocks())
    assert np.array_equal(o.states(), g.states())
    assert np.array_equal(bits(o.theta()), bits(g.theta()))
    to, oo, so, qo, _ = o.counts()
    tg, og, sg, qg, _ = g.counts()
    assert np.array_equal(to, tg) and np.array_equal(oo, og)
    assert np.array_equal(bits(so), bits(sg)) and np.array_equal(bits(qo), bits(qg))
    st, bl = g.states(), g.blocks()
    exact = np.bincount(st, weights=np.diff(bl).astype(np.float64), minlength=K).astype(np.uint64)
    assert exact.max() > (1 << 24) and not np.array_equal(exact, og)        # the rounding is really there
    assert int(exact.sum()) == T
