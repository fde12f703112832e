"""Size-independent properties at BASELINE.json's full size (10^8 positions, 5 states, dynamic blocks), where the
CPU checker is too slow to run inside the test suite: the block structure against an independent evaluation of
its definition, conservation laws of the count pass and the marginals, run-to-run determinism, and equality of
the two block-enumeration paths (group summary / float stream)."""
import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

T, K = 100_000_000, 5


@pytest.fixture(scope="module")
def big_trace():
    return ol.trace(T, K, 3)


def make_chain(hml, x, summary, seed=1):
    c = hml.Chain(device=0, seed=seed)
    c.set_option("weight_keys", 1 if summary else 0)
    c.load(x)
    c.set_model(K, c.autoprior(0.2, 0.9))
    c.sample_prior()
    return c


def test_full_size_properties(hml, big_trace):
    x = big_trace
    c = make_chain(hml, x, summary=True)
    c.iterate("F", 30, 0)
    c.iterate("F", 20, 4)      # 5 recorded sweeps
    c.sync()
    st = c.stats()
    assert st["sweeps"] == 50
    # --- the block structure is exactly {0} u {t : !(w[t] < thr)} for the threshold of the LAST sweep ...
    starts = c.blocks()
    B = len(starts) - 1
    assert starts[0] == 0 and starts[-1] == T and np.all(np.diff(starts.astype(np.int64)) > 0)
    trans, occ, sx, sq, n = c.counts()
    # ... whose threshold was set by the parameters BEFORE the last resampling; re-derive it from a fresh
    # enumeration at an explicit threshold instead: both enumeration paths must agree with numpy on the weights
    w = c.weights()
    for thr in (0.5, 1.7, 3.0):
        c.create_blocks(thr)
        got = c.blocks()
        flags = ~(w < np.float32(thr))
        flags[0] = True
        expect = np.flatnonzero(flags)
        assert len(got) - 1 == len(expect) and np.array_equal(got[:-1], expect.astype(np.uint32)), thr
        s1, s2 = c.block_stats()
        # sums of blocks are consistent with the data to float accuracy (integral-array differences)
        tot = np.add.reduceat(x.astype(np.float64), expect)
        assert np.allclose(s1, tot, rtol=0, atol=2e-3 * np.sqrt(np.diff(np.append(expect, T))) + 6.0)
    # --- conservation in the count pass of the last sweep: every position is counted exactly once
    assert int(occ.sum()) == T
    assert int(trans.sum()) == T          # (N-1) self transitions + 1 entering transition per block
    assert int(np.trace(trans)) >= T - 2 * B
    # --- marginals: segments tile [0,T), every position has exactly #recorded counts
    seg, cnt = c.marginals_rle()
    assert int(seg.sum()) == T
    assert np.all(cnt.sum(1) == 5)
    assert c.recorded_sweeps() == 5
    # compression in the regime the survey measured for this generator (B ~ 1.8e5)
    assert 1.2e5 < B < 2.6e5
    assert st["forward_serial"] <= 64
    c.close()


def test_summary_and_float_stream_give_identical_chains(hml, big_trace):
    """the fused summary kernel (default) against scan + scatter + statistics over the float weights"""
    x = big_trace
    res = []
    for summary in (True, False):
        c = make_chain(hml, x, summary=summary, seed=5)
        c.iterate("F", 25, 5)
        c.sync()
        res.append((c.blocks(), c.states(), c.theta(), c.transitions()[0], c.marginals_rle(), c.stats(), c.block_stats()))
        c.close()
    a, b = res
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
    assert np.array_equal(a[4][0], b[4][0]) and np.array_equal(a[4][1], b[4][1])
    assert a[5]["block_updates"] == b[5]["block_updates"]
    assert np.array_equal(a[6][0].view(np.uint32), b[6][0].view(np.uint32))
    assert np.array_equal(a[6][1].view(np.uint32), b[6][1].view(np.uint32))


def test_compat_chain_at_full_size_is_the_reference_chain(hml, big_trace):
    """The reference-compatible mode at BASELINE config 3's FULL size (10^8 positions, 5 states, dynamic blocks): eight sweeps
    of a compat chain against the checker's REFERENCE mode (sequential mt19937, glibc arithmetic, float Kahan sums in block
    order, `size_t += float` counts - the mode that reproduces the unmodified reference binary's files byte for byte,
    also by hand at this size: DESIGN.md 2) - block structure, state sequence, parameter bits, transition matrix, and the
    counts, which exceed 2^24 here and therefore ROUND in the reference (src/StateSequence/ForwardBackward.hpp:183-187)."""
    x = big_trace
    seed = 9
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
    for n in (3, 5):
        o.iterate("F", n, 0)
        g.iterate("F", n, 0)
        g.sync()
        assert np.array_equal(o.blocks(), g.blocks())
        assert np.array_equal(o.states(), g.states())
        assert np.array_equal(o.theta().view(np.uint32), g.theta().view(np.uint32))
        Ao, pio = o.transitions()
        Ag, pig = g.transitions()
        assert np.array_equal(Ao.view(np.uint32), Ag.view(np.uint32)) and np.array_equal(pio.view(np.uint32), pig.view(np.uint32))
        to, oo, so, qo, _ = o.counts()
        tg, og, sg, qg, _ = g.counts()
        assert np.array_equal(to, tg) and np.array_equal(oo, og)
        assert np.array_equal(so.view(np.uint32), sg.view(np.uint32)) and np.array_equal(qo.view(np.uint32), qg.view(np.uint32))
    assert int(og.max()) > (1 << 24)   # the regime in which the reference's counts round
    g.close()
    o.close()
