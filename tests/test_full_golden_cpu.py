"""The CPU checker in reference mode against files the UNMODIFIED reference binary wrote at BASELINE.json's FULL sizes
(configs 2, 3 and 4: 10^7 and 10^8 positions; tests/golden/full/, tests/golden/make_full_golden.py): byte for byte.
This pins the checker where the small fixtures cannot: per-state counts above 2^24, which the reference accumulates as
`size_t += float` and therefore rounds (src/StateSequence/ForwardBackward.hpp:183-187), ~1526 cells of the integral array
(src/Statistics/IntegralArray.hpp:136-191), blocks longer than the uint16 pointer range (src/Blocks/BreakpointArray.hpp:130-184),
marginals of 10^5-block sweeps (src/StateMarginals.hpp:51-137,268-310).  About a minute per 10^8-position case; skipped
where memory does not allow it or with HML_SKIP_FULLSIZE=1."""
import os
import subprocess
import tempfile

import pytest

from tests import full_golden_util as fg
from tests import oracle_lib as ol

ORACLE_CLI = os.path.join(ol.ORACLE_DIR, "hammlet_oracle")


@pytest.mark.parametrize("case", sorted(fg.MANIFEST))
def test_reference_mode_reproduces_full_size_reference_files(case):
    if os.environ.get("HML_SKIP_FULLSIZE") == "1":
        pytest.skip("HML_SKIP_FULLSIZE=1")
    m = fg.MANIFEST[case]
    if m.get("reference_seconds", 0) > 600 and os.environ.get("HML_FULLSIZE_SLOW") != "1":
        # config 4's chain: 50 sweeps of 10^8 blocks and ten states (70 minutes in the reference binary, about as long in the
        # checker); the GPU reproduces its files in tests/test_gpu_full_golden.py
        pytest.skip("a %d-second run of the reference binary: set HML_FULLSIZE_SLOW=1" % m["reference_seconds"])
    K = int(m["flags"].split()[1])
    if not fg.enough_memory(case, 40 + 8 * K):
        pytest.skip("not enough memory for %d positions" % m["T"])
    ol.load()
    x = fg.trace(case)
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "in.f32")
        x.tofile(raw)
        del x
        r = subprocess.run([ORACLE_CLI, "--raw", raw, "-o", os.path.join(tmp, "o-"), ".csv", "-a"] + m["flags"].split() + ["-O"] + m["outputs"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for o in m["outputs"]:
            assert fg.matches_golden(case, o, os.path.join(tmp, "o-%s.csv" % o)), (case, o)
