"""mapLinesToGenome / combineCounts (hammlet_amd/csrc/host/*_main.cpp) against the reference's own tools:
tests/golden/genome_tools/expected.json holds, per case, the command line, the standard output, whether the tool
succeeded, the message of the exception it ended with, and the text of every file it wrote - all produced by the
reference binaries (tests/golden/make_genome_tools_golden.py).  Host tools: no GPU involved."""
import gzip
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden", "genome_tools")
with open(os.path.join(GOLD, "expected.json")) as f:
    EXPECTED = json.load(f)
CASES = [(tool, name) for tool in sorted(EXPECTED) for name in sorted(EXPECTED[tool])]


@pytest.fixture(scope="module")
def tools():
    from hammlet_amd import build
    paths = {t: os.path.join(build.PKG_DIR, t) for t in build.GENOME_TOOLS}
    if not all(os.path.exists(p) for p in paths.values()):
        build.build_cli(force=False, verbose=False)
    return paths


def files_of(d):
    out = {}
    for name in sorted(os.listdir(d)):
        raw = open(os.path.join(d, name), "rb").read()
        out[name] = (gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw).decode("latin-1")
    return out


@pytest.mark.parametrize("tool,name", CASES)
def test_same_output_as_the_reference_tool(tools, tmp_path, tool, name):
    want = EXPECTED[tool][name]
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    argv = [a.replace("@OUT@", out_dir) for a in want["argv"]]
    stdin = open(os.path.join(GOLD, "inputs", want["stdin"])) if want["stdin"] else subprocess.DEVNULL
    r = subprocess.run([tools[tool]] + argv, cwd=os.path.join(GOLD, "inputs"), stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=60)
    assert r.stdout.decode("latin-1").replace(out_dir, "@OUT@") == want["stdout"]
    assert (r.returncode == 0) == want["ok"], r.stderr
    if want["message"] is not None:
        # the reference aborts with the exception's text; the tool prints "TOOL: text" and exits with status 1
        assert r.returncode == 1
        assert r.stderr.decode("latin-1").strip() == "%s: %s" % (tool, want["message"])
    assert files_of(out_dir) == want["files"]


def test_gzip_output_is_a_gzip_stream(tools, tmp_path):
    """the position / count files are real gzip files (the reference writes them through zlib's gzopen)"""
    out = str(tmp_path / "sum")
    subprocess.run([tools["combineCounts"], "-i", "+", "c7", "c8", "-o", out], cwd=os.path.join(GOLD, "inputs"), check=True,
                   stdout=subprocess.DEVNULL)
    for suffix in ("-pos.csv.gz", "-count.csv.gz"):
        raw = open(out + suffix, "rb").read()
        assert raw[:2] == b"\x1f\x8b"
        assert len(gzip.decompress(raw).splitlines()) == sum(int(l.split("\t")[1]) for l in open(out + "-size.csv"))
