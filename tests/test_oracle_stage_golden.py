"""The CPU checker against stage-level golden vectors of the UNMODIFIED reference (tests/golden/stages/*.npz, produced
by tests/golden/make_stage_golden.py from oracle/ref_harness.cpp + the reference's own headers): maxlet coefficients,
noise estimate, breakpoint weights, integral array, auto prior, and per threshold the block list, block statistics
and emission terms - bit for bit (arrays above 8192 elements through their SHA-256)."""
import glob
import hashlib
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

STAGES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stages", "*.npz")))
MEANS_VARS = np.array([-1.0, 0.04, 0.25, 0.09, 1.5, 0.5], np.float32)   # ref_harness.cpp's fixed parameters


def same(fx, key, got):
    got = np.ascontiguousarray(got)
    if key in fx:
        want = fx[key]
        assert want.shape == got.shape, (key, want.shape, got.shape)
        assert want.tobytes() == got.astype(want.dtype).tobytes(), key
    else:
        assert int(fx[key + "_size"]) == got.size, key
        assert fx[key + "_sha256"].tobytes() == hashlib.sha256(got.tobytes()).digest(), key


def test_fixtures_present():
    assert len(STAGES) >= 10


@pytest.mark.parametrize("path", STAGES, ids=[os.path.basename(p)[:-4] for p in STAGES])
def test_checker_matches_reference_stage_by_stage(path):
    fx = np.load(path)
    T, levels, seed, mult = int(fx["T"]), int(fx["levels"]), int(fx["seed"]), float(fx["mult"])
    x = ol.trace(T, levels, seed)
    o = ol.OracleChain(K=3, weight_mult=mult)
    o.load(x)
    same(fx, "coeffs", o.coeffs())
    assert np.float64(o.sigma_hat()).tobytes() == fx["sigma"].tobytes()
    same(fx, "weights", o.weights())
    a, b = o.integral()
    same(fx, "integral", np.stack([a, b], 1).reshape(-1))
    same(fx, "autoprior", o.autoprior())
    o.init_model()
    o.set_params(MEANS_VARS, np.full((3, 3), 1.0 / 3, np.float32), np.full(3, 1.0 / 3, np.float32))
    for k, thr in enumerate(fx["thresholds"]):
        o.enumerate_blocks(float(thr))
        same(fx, "starts_%d" % k, o.blocks())
        s1, s2 = o.block_stats()
        same(fx, "sum_%d" % k, s1)
        same(fx, "sumsq_%d" % k, s2)
        same(fx, "E_%d" % k, o.emission_terms().reshape(-1))
    o.close()
