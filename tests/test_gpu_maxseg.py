"""Device maximum-posterior-margin segmentation (hml_max_segmentation, `-O X`) against the restated reference tool
(reference src/tools/maxSegmentation.cpp:53-82; the restatement is pinned on the reference tool's own outputs by
tests/test_maxseg_cpu.py) applied to the marginals of the same chain."""
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def tool_text(run_len, run_state):
    """what the tool prints for these runs (its running state starts at 0)"""
    lines = []
    if len(run_state) and run_state[0] != 0:
        lines.append("0\t0\n")
    lines += ["%d\t%d\n" % (l, s) for l, s in zip(run_len, run_state)]
    return "".join(lines)


@pytest.mark.parametrize("T,K,seed,scheme", [(100000, 3, 1, [("F", 60, 1)]), (20000, 4, 3, [("M", 20, 2), ("F", 40, 3)]),
                                             (300000, 5, 7, [("F", 30, 5)]), (1000, 6, 2, [("F", 50, 1)]), (16, 2, 5, [("F", 20, 1)]),
                                             (70000, 3, 9, [("F", 10, 0)])])
def test_device_segmentation_equals_the_tool_on_the_marginals(hml, T, K, seed, scheme):
    x = ol.trace(T, min(K, 5), seed)
    g = hml.Chain(device=0, seed=seed)
    g.load(x)
    g.set_model(K, g.autoprior(0.2, 0.9))
    g.sample_prior()
    for method, iters, thin in scheme:
        g.iterate(method, iters, thin)
    g.sync()
    seg, cnt = g.marginals_rle()
    want = ol.max_segmentation_text(hml.marginals_text(seg, cnt))
    run_len, run_state = g.max_segmentation()
    assert int(run_len.sum()) == T
    assert np.all(run_state[1:] != run_state[:-1])
    assert tool_text(run_len, run_state) == want


def test_many_marginal_segments(hml):
    """uncompressed trace (every position its own block): 10^5-10^6 marginal segments, several scan chunks"""
    x = ol.trace(400000, 3, 4)
    g = hml.Chain(device=0, seed=3)
    g.load(x)
    g.scale_weights(1e9)
    g.set_model(3, g.autoprior(0.2, 0.9))
    g.sample_prior()
    g.iterate("F", 12, 1)
    g.sync()
    seg, cnt = g.marginals_rle()
    assert seg.size > 10000
    run_len, run_state = g.max_segmentation()
    assert tool_text(run_len, run_state) == ol.max_segmentation_text(hml.marginals_text(seg, cnt))


def test_cli_maxsegmentation_file_is_the_tool_output_for_the_marginals_file(hml, tmp_path):
    from hammlet_amd import build
    x = ol.trace(50000, 3, 21)
    raw = tmp_path / "in.f32"
    x.tofile(raw)
    subprocess.run([build.CLI_PATH, "-raw", str(raw), "-a", "-s", "4", "-R", "6", "-i", "F", "40", "2", "-O", "M", "X", "-w",
                    "-o", str(tmp_path / "r-"), ".csv"], check=True)
    tool = subprocess.run([build.TOOL_PATH, "-i", str(tmp_path / "r-marginals.csv")], check=True, capture_output=True, text=True).stdout
    assert open(tmp_path / "r-maxsegmentation.csv").read() == tool
    # X alone records marginals on the device too
    subprocess.run([build.CLI_PATH, "-raw", str(raw), "-a", "-s", "4", "-R", "6", "-i", "F", "40", "2", "-O", "X", "-w",
                    "-o", str(tmp_path / "s-"), ".csv"], check=True)
    assert open(tmp_path / "s-maxsegmentation.csv").read() == tool
    assert not os.path.exists(tmp_path / "s-marginals.csv")
