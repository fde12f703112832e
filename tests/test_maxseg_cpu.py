"""maxSegmentation: the host tool and the checker's restatement against the golden outputs of the reference's own tool
(reference src/tools/maxSegmentation.cpp; goldens by tests/golden/make_maxseg_golden.py)."""
import glob
import os
import subprocess

import pytest

from tests import oracle_lib as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PAIRS = [(p, p[:-len("marginals.csv")] + "maxsegmentation.txt") for p in sorted(glob.glob(os.path.join(GOLD, "*", "marginals.csv")))]
PAIRS += [(p, p[:-len(".marginals")] + ".maxseg") for p in sorted(glob.glob(os.path.join(GOLD, "maxseg", "*.marginals")))]


def test_there_are_goldens():
    assert len(PAIRS) >= 20 and all(os.path.exists(b) for _, b in PAIRS)


@pytest.mark.parametrize("inp,want", PAIRS, ids=[os.path.relpath(a, GOLD) for a, _ in PAIRS])
def test_restatement_and_host_tool_match_the_reference_tool(inp, want):
    from hammlet_amd import build
    build.build_cli()
    expected = open(want).read()
    assert ol.max_segmentation_text(open(inp).read()) == expected
    got = subprocess.run([build.TOOL_PATH, "-i", inp], check=True, capture_output=True, text=True).stdout
    assert got == expected
    got = subprocess.run([build.TOOL_PATH], check=True, capture_output=True, text=True, stdin=open(inp)).stdout
    assert got == expected


SORT_PAIRS = [(p, os.path.join(os.path.dirname(p), "sortstates.txt")) for p in sorted(glob.glob(os.path.join(GOLD, "*", "parameters.csv")))]


@pytest.mark.parametrize("inp,want", SORT_PAIRS, ids=[os.path.relpath(a, GOLD) for a, _ in SORT_PAIRS])
def test_sort_states_tool_matches_the_reference_script(inp, want):
    """hammlet_amd/sortStates against the output of the reference's bin/sortStates (tests/golden/make_sortstates_golden.py)"""
    from hammlet_amd import build
    build.build_cli()
    assert len(SORT_PAIRS) >= 8
    got = subprocess.run([build.SORT_TOOL_PATH, inp], check=True, capture_output=True, text=True).stdout
    assert got == open(want).read()
