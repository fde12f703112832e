"""GPU text reader (hml_text_*, through the C ABI) against the golden values of the reference's own reader and,
on seeded random text, against the checker's restatement of it (`while ( input >> v )`, reference
src/wavelet.hpp:131): bit-identical values, and the same place to stop."""
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as ol
from tests.test_text_reader_cpu import CASES, golden, random_tokens, stops_early

pytestmark = pytest.mark.gpu


def same(a, b):
    return a.size == b.size and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name", CASES)
def test_reader_matches_the_reference_reader(hml, name):
    text, want = golden(name)
    for chunk, feed in ((0, None), (512, 97), (4096, 4096), (5000, 1), (65536, 30000)):
        if feed == 1 and len(text) > 40000:
            continue
        got, info = hml.parse_text(text, chunk_bytes=chunk, feed_bytes=feed, with_info=True)
        assert same(got, want), (name, chunk, feed, got.size, want.size)
        assert info["stopped"] == stops_early(name)
        assert info["bytes"] == len(text)


def test_reader_on_random_tokens_including_irregular_ones(hml):
    rng = np.random.default_rng(11)
    toks = [t for t in random_tokens(rng, 300000) if t not in ("5e", ".", "-", "abc", "1,5", "0x1p3", "nan", "1e39")]
    finite = np.isfinite(ol.parse_tokens(toks)[2])          # an overflowing value ends the reference's loop
    toks = [t for t, ok in zip(toks, finite) if ok]
    seps = np.array([" ", "\n", "\t", "  ", "\r\n", " \n "])[rng.integers(0, 6, len(toks))]
    text = "".join(t + s for t, s in zip(toks, seps)).encode()
    want, stopped = ol.parse_text(text)
    assert not stopped and want.size >= len(toks)
    for chunk in (0, 1 << 16, 12345):
        got, info = hml.parse_text(text, chunk_bytes=chunk, with_info=True)
        assert same(got, want)
        assert not info["stopped"] and info["irregular_tokens"] > 0 and info["host_chunks"] == 0


def test_reader_stops_where_the_reference_stops_in_a_later_chunk(hml):
    rng = np.random.default_rng(5)
    vals = rng.normal(size=50000)
    lines = ["%.5f" % v for v in vals]
    lines[33333] = "NA"
    text = ("\n".join(lines) + "\n").encode()
    want, stopped = ol.parse_text(text)
    assert stopped and want.size == 33333
    for chunk in (0, 4096, 100000):
        got, info = hml.parse_text(text, chunk_bytes=chunk, with_info=True)
        assert same(got, want) and info["stopped"]


def test_chunk_with_more_irregular_tokens_than_the_list_holds(hml):
    # 70000 21-digit tokens in one chunk: beyond the 65536-entry list, the chunk goes through the stream extraction whole
    toks = ["1.%020d" % (i * 7919) for i in range(70000)]
    text = (" ".join(toks) + "\n1.5\n").encode()
    want, _ = ol.parse_text(text)
    got, info = hml.parse_text(text, with_info=True)
    assert same(got, want) and info["host_chunks"] == 1
    got, info = hml.parse_text(text, chunk_bytes=65536, with_info=True)
    assert same(got, want) and info["host_chunks"] == 0 and info["irregular_tokens"] == 70000


def test_token_longer_than_the_staging_buffer_is_an_error(hml):
    with pytest.raises(hml.HmlError):
        hml.parse_text(b"1 " + b"7" * 2000 + b" 2", chunk_bytes=1024)


def test_file_source_and_full_size_column(hml, tmp_path):
    """a 2*10^6-value column through the file interface (pinned staging buffer, read() straight into it)"""
    rng = np.random.default_rng(3)
    x = (rng.normal(size=2_000_000) * 0.3 + np.repeat(rng.integers(-2, 3, 4000), 500)).astype(np.float32)
    p = tmp_path / "col.txt"
    with open(p, "w") as f:
        f.write("\n".join("%.9g" % v for v in x))
    got, info = hml.parse_text(str(p), with_info=True)
    assert same(got, x)          # %.9g round-trips float32
    got = hml.parse_text(str(p), chunk_bytes=1 << 20)
    assert same(got, x)


def test_cli_reads_text_through_the_gpu_reader(hml, tmp_path):
    """the driver's -f path: same files as with the float32 extension input"""
    from hammlet_amd import build
    x = ol.trace(30000, 3, 77)
    txt, raw = tmp_path / "in.txt", tmp_path / "in.f32"
    with open(txt, "w") as f:
        f.write("\n".join("%.9g" % v for v in x) + "\n")
    x.tofile(raw)
    for tag, args in (("a", ["-f", str(txt)]), ("b", ["-raw", str(raw)])):
        subprocess.run([build.CLI_PATH] + args + ["-a", "-R", "5", "-s", "3", "-i", "F", "30", "1", "-O", "M", "S", "P", "-w",
                                                   "-o", str(tmp_path / (tag + "-")), ".csv"], check=True)
    for kind in ("marginals", "sequences", "parameters"):
        assert open(tmp_path / ("a-%s.csv" % kind)).read() == open(tmp_path / ("b-%s.csv" % kind)).read()


def test_cli_reads_standard_input_and_concatenates_files(hml, tmp_path):
    """no -f: the text comes from standard input (reference src/main.cpp:284-289); several -f files are read one after
    the other (the man page's contract)"""
    from hammlet_amd import build
    x = ol.trace(24000, 3, 5)
    whole, a, b = tmp_path / "whole.txt", tmp_path / "a.txt", tmp_path / "b.txt"
    lines = ["%.9g" % v for v in x]
    whole.write_text("\n".join(lines) + "\n")
    a.write_text("\n".join(lines[:10000]))          # no newline at the end of the first file
    b.write_text("\n".join(lines[10000:]) + "\n")
    common = ["-a", "-R", "3", "-s", "3", "-i", "F", "20", "1", "-O", "M", "P", "-w"]
    subprocess.run([build.CLI_PATH, "-f", str(whole)] + common + ["-o", str(tmp_path / "w-"), ".csv"], check=True)
    with open(whole, "rb") as f:
        subprocess.run([build.CLI_PATH] + common + ["-o", str(tmp_path / "s-"), ".csv"], check=True, stdin=f)
    subprocess.run([build.CLI_PATH, "-f", str(a), str(b)] + common + ["-o", str(tmp_path / "c-"), ".csv"], check=True)
    for kind in ("marginals", "parameters"):
        want = open(tmp_path / ("w-%s.csv" % kind)).read()
        assert open(tmp_path / ("s-%s.csv" % kind)).read() == want
        assert open(tmp_path / ("c-%s.csv" % kind)).read() == want


def test_avg_tool_matches_the_reference_tool(hml):
    """hammlet_amd/avg (values through the GPU text reader, window sums in the reference's order) against the outputs of
    the reference's src/tools/avg.cpp (tests/golden/avg, made by tests/golden/make_avg_golden.py)"""
    import glob
    from hammlet_amd import build
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    outs = sorted(glob.glob(os.path.join(gold, "avg", "*.out")))
    assert len(outs) >= 10
    for path in outs:
        name, w = os.path.basename(path)[:-4].rsplit(".w", 1)
        with open(os.path.join(gold, "text", name + ".txt"), "rb") as f:
            got = subprocess.run([build.AVG_TOOL_PATH, w], stdin=f, check=True, capture_output=True).stdout
        assert got == open(path, "rb").read(), (name, w)
