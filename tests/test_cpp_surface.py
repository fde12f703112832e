"""The in-process C++ surface: tests/fixtures/reference_shaped_driver.cpp builds and runs the sampler
with the call shapes of the reference's src/main.cpp - `MaxletTransform(fin, inputValues, stats, nrDataDim)`,
`HaarBreakpointWeights(inputValues)`, `Statistics(stats, nrDataDim)`, `Blocks(inputValues)`, `Theta(tau_theta,
nrDataDim, mappingType, RNG)`, `theta.sample / pi.sample / A.sample`, `y.createBlocks(theta)`, `sampleHMM(...)`, and a
hand-written loop over `StateSequence::sample(...eleven arguments...)` - and must compile unedited against
include/hammlet (CPU) and reproduce the `hammlet` driver's files (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as ol

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "fixtures", "reference_shaped_driver.cpp")
CLI = os.path.join(REPO, "hammlet_amd", "hammlet")
OUTS = ["marginals", "sequences", "parameters", "blocks", "compression"]


def build_driver(out):
    from hammlet_amd import build
    build.build_library()
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-I", os.path.join(REPO, "include"), "-o", out, SRC,
           "-L", os.path.join(REPO, "hammlet_amd"), "-lhammlet_hip", "-Wl,-rpath," + os.path.join(REPO, "hammlet_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_reference_shaped_driver_compiles_and_links(tmp_path):
    exe = build_driver(str(tmp_path / "driver"))
    # without a GPU the first device call fails loudly, in the reference's error format
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "none.txt", "p-", ".csv", "1", "3", "1", "1", "F", "1", "1"], capture_output=True, text=True)
        assert r.returncode == 1 and "[ERROR]" in r.stderr and "Terminating HaMMLET. The rest is silence." in r.stderr


def test_integration_document_shows_the_fixture():
    """INTEGRATION.md section B is this driver's text: every statement of the fixture's main() that touches the surface
    appears there verbatim"""
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    for line in open(SRC).read().splitlines():
        t = line.strip()
        if t.startswith("//"):
            continue
        if any(k in t for k in ("MaxletTransform(", "HaarBreakpointWeights(", "S ia(", "B waveletBlocks(", "Emissions<S, B> y(",
                                "Theta<NormalInverseGamma> theta(", "q.sample(", "sampleHMM(y, q,", "records.record(theta)")):
            assert t in doc, t


@pytest.mark.gpu
@pytest.mark.parametrize("K,selftrans,mult,scheme", [
    (3, 1, 1.0, "F 30 3"),
    (3, 1, 1.0, "f 30 3"),
    (4, 0, 1.5, "M 10 2 S P f 12 2 D F 8 1 m 4 1"),
    (5, 1, 1.0, "m 6 0 S P f 10 0 D f 12 3"),
])
def test_reference_shaped_driver_reproduces_the_cli(tmp_path, K, selftrans, mult, scheme):
    exe = build_driver(str(tmp_path / "driver"))
    x = ol.trace(60000, K, 17)
    txt = str(tmp_path / "in.txt")
    np.savetxt(txt, x, fmt="%.9g")
    r = subprocess.run([exe, txt, str(tmp_path / "d-"), ".csv", "7", str(K), str(selftrans), repr(mult)] + scheme.split(),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    flags = ["-f", txt, "-o", str(tmp_path / "c-"), ".csv", "-w", "-a", "-s", str(K), "-R", "7", "-m", repr(mult)]
    if not selftrans:
        flags.append("-S")
    c = subprocess.run([CLI] + flags + ["-i"] + scheme.upper().split() + ["-O"] + OUTS, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr
    for name in OUTS:
        a = open(str(tmp_path / ("d-%s.csv" % name))).read()
        b = open(str(tmp_path / ("c-%s.csv" % name))).read()
        assert a == b and len(a) > 0, name
