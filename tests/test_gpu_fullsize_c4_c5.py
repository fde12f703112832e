"""Size-independent properties at the full sizes of BASELINE.json's configs 4 and 5, where the CPU checker is too slow
to run inside the suite: C4 = 10^8 positions, 10 states (one of its 8 chains per GPU); C5 = 2.5*10^8
simulated read-depth positions, 5-state model, compression ~1.5 (1.7*10^8 blocks per sweep: the dense forward
geometry, the float weight stream, the two-level backward chain).  Checked: the block structure against its
definition evaluated by numpy, the conservation laws of the count pass and of the marginals, equality of the chains
that the alternative enumeration paths / forward geometries produce, bounded repair work, chain ids."""
import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def run_chain(hml, x, K, chain_id, sweeps, keys=None, seed=1):
    c = hml.Chain(device=0, seed=seed, chain_id=chain_id)
    if keys is not None:
        c.set_option("weight_keys", keys)
    c.load(x)
    c.set_model(K, c.autoprior(0.2, 0.9))
    c.sample_prior()
    for n, thin in sweeps:
        c.iterate("F", n, thin)
    c.sync()
    return c


def snapshot(c):
    return {"blocks": c.blocks(), "states": c.states(), "theta": c.theta(), "A": c.transitions()[0], "marg": c.marginals_rle(),
            "stats": c.stats(), "counts": c.counts()}


def assert_same_chain(a, b):
    assert np.array_equal(a["blocks"], b["blocks"]) and np.array_equal(a["states"], b["states"])
    assert np.array_equal(bits(a["theta"]), bits(b["theta"])) and np.array_equal(bits(a["A"]), bits(b["A"]))
    assert np.array_equal(a["marg"][0], b["marg"][0]) and np.array_equal(a["marg"][1], b["marg"][1])
    assert a["stats"]["block_updates"] == b["stats"]["block_updates"]
    for u, v in zip(a["counts"], b["counts"]):
        assert np.array_equal(np.asarray(u).view(np.uint8), np.asarray(v).view(np.uint8))


def check_conservation(c, T, recorded):
    trans, occ, sx, sq, n = c.counts()
    B = c.num_blocks()
    assert int(occ.sum()) == T                      # every position is counted exactly once
    assert int(trans.sum()) == T                    # N - 1 self transitions + 1 entering transition per block
    assert int(np.trace(trans)) >= T - 2 * B
    starts = c.blocks()
    assert starts[0] == 0 and starts[-1] == T and len(starts) == B + 1
    assert np.all(np.diff(starts.astype(np.int64)) > 0)
    q = c.states()
    # the count pass against numpy on the state sequence: occupancies and transitions between different states
    sizes = np.diff(starts.astype(np.int64))
    K = len(occ)
    assert np.array_equal(np.bincount(q, weights=sizes, minlength=K).astype(np.int64), occ.astype(np.int64))
    prev = np.concatenate([[0], q[:-1]]).astype(np.int64)
    off = np.bincount(prev * K + q, minlength=K * K).reshape(K, K)
    assert np.array_equal(off + np.diag(np.bincount(q, weights=sizes - 1, minlength=K).astype(np.int64)), trans.astype(np.int64))
    seg, cnt = c.marginals_rle()
    assert int(seg.sum()) == T and np.all(cnt.sum(1) == recorded) and c.recorded_sweeps() == recorded
    return B


def check_block_definition(c, x, thresholds):
    w = c.weights()
    T = len(w)
    for thr in thresholds:
        c.create_blocks(thr)
        got = c.blocks()
        flags = ~(w < np.float32(thr))
        flags[0] = True
        expect = np.flatnonzero(flags)
        assert len(got) - 1 == len(expect) and np.array_equal(got[:-1], expect.astype(np.uint32)), thr
        s1, _ = c.block_stats()
        tot = np.add.reduceat(x.astype(np.float64), expect)
        assert np.allclose(s1, tot, rtol=2e-5, atol=2e-3 * np.sqrt(np.diff(np.append(expect, T))) + 6.0), thr


# ---------------------------------------------------------------------------------------------- C4
@pytest.fixture(scope="module")
def c4_trace():
    return ol.trace(100_000_000, 10, 4)


def test_config4_full_size_properties(hml, c4_trace):
    T, K = 100_000_000, 10
    x = c4_trace
    c = run_chain(hml, x, K, chain_id=3, sweeps=[(60, 0), (20, 4)])
    st = c.stats()
    assert st["sweeps"] == 80
    B = check_conservation(c, T, recorded=5)
    assert 1.0e5 < B < 4.0e5                                   # the compression regime of this generator
    # repair work stays bounded: unused states are twins while the chain burns in, the
    # sequential finisher must stay an exception
    assert st["forward_refits"] <= 2_000_000 and st["forward_serial"] <= 50_000, st
    base = snapshot(c)
    check_block_definition(c, x, (0.5, 1.7, 3.0))
    c.close()
    # the float weight stream gives the same chain as the group summary
    f = run_chain(hml, x, K, chain_id=3, sweeps=[(60, 0), (20, 4)], keys=0)
    assert_same_chain(base, snapshot(f))
    f.close()
    # another chain id of the same seed (config 4 runs 8 of them): an independent chain, same invariants
    d = run_chain(hml, x, K, chain_id=5, sweeps=[(40, 0), (10, 2)])
    check_conservation(d, T, recorded=5)
    assert not np.array_equal(bits(d.theta()), bits(base["theta"]))
    d.close()


# ---------------------------------------------------------------------------------------------- C5
@pytest.fixture(scope="module")
def c5_trace():
    return ol.synth_depth(250_000_000, depth=15.0, ln_sigma=0.15, seed=5, nthreads=16)


def test_config5_full_size_properties(hml, c5_trace, monkeypatch):
    T, K = 250_000_000, 5
    x = c5_trace
    sweeps = [(10, 0), (6, 2)]
    c = run_chain(hml, x, K, chain_id=0, sweeps=sweeps)
    st = c.stats()
    B = check_conservation(c, T, recorded=3)
    assert B > T // 4                                           # weakly compressed: the dense geometry is in use
    assert st["forward_serial"] <= 100_000, st
    base = snapshot(c)
    check_block_definition(c, x, (2.0, 40.0))
    c.close()
    # the sparse forward geometry (chunks of 4, one-workgroup chain) gives the same chain, bit for bit
    monkeypatch.setenv("HML_DENSE_MIN_BLOCKS", "4000000000")
    s = run_chain(hml, x, K, chain_id=0, sweeps=sweeps)
    assert_same_chain(base, snapshot(s))
    s.close()
    monkeypatch.delenv("HML_DENSE_MIN_BLOCKS")
    # ... and so does the float weight stream from the first sweep on
    f = run_chain(hml, x, K, chain_id=0, sweeps=sweeps, keys=0)
    assert_same_chain(base, snapshot(f))
    f.close()
