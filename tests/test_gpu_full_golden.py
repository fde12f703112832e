"""`hammlet -compat` on the GPU against files the UNMODIFIED reference binary wrote at BASELINE.json's FULL sizes (configs 2, 3
and 4: 10^7 and 10^8 positions, tests/golden/full/): marginals, parameters and compression byte for byte - the reference's
chain at the reference's seed, at the sizes the headline is quoted on (north star: "state-marginal counts at a fixed RNG
seed").  Reference: src/StateSequence/ForwardBackward.hpp:170-200 (counts that round above 2^24), src/StateMarginals.hpp:268-310."""
import os
import subprocess
import tempfile

import pytest

from tests import full_golden_util as fg

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "hammlet_amd", "hammlet")


@pytest.mark.parametrize("case", sorted(fg.MANIFEST))
def test_cli_compat_writes_the_full_size_reference_files(case):
    m = fg.MANIFEST[case]
    x = fg.trace(case)
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "in.f32")
        x.tofile(raw)
        del x
        r = subprocess.run([CLI, "-compat", "-raw", raw, "-o", os.path.join(tmp, "g-"), ".csv", "-a"] + m["flags"].split() + ["-O"] + m["outputs"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for o in m["outputs"]:
            assert fg.matches_golden(case, o, os.path.join(tmp, "g-%s.csv" % o)), (case, o)
