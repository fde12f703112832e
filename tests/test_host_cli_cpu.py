"""Host-side logic of the `hammlet` driver that runs before any GPU call: flag parsing and its error
messages (reference src/Parser.hpp:163-193, src/main.cpp:33-65), checked without a GPU."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "hammlet_amd", "hammlet")
GOLD = os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="module")
def cli():
    from hammlet_amd import build
    build.build_cli()
    return CLI


def run(cli, *args, stdin=""):
    return subprocess.run([cli] + list(args), input=stdin, capture_output=True, text=True)


def test_arguments_dump_matches_reference(cli):
    """`-g`: one line per flag group, "[*] flags : tokens" - the reference's own output for the same command
    line (tests/golden/cli_g_output.txt); the driver's extension flags follow after the reference's."""
    r = run(cli, "-g", "-a", "-s", "4", "-R", "1", "-i", "F", "1", "0", "-f", "tiny.txt", "-o", "out-", ".csv", "-w",
            "-O", "marginals", "blocks", "-t", "0.3", "0.7")
    got = r.stdout.splitlines()
    want = open(os.path.join(GOLD, "cli_g_output.txt")).read().splitlines()
    assert got[:len(want)] == want
    assert got[len(want):len(want) + 3] == ["[ ] -raw :", "[ ] -device : 0", "[ ] -chain : 0"]


def test_parser_errors(cli):
    tail = "\nTerminating HaMMLET. The rest is silence.\n"
    r = run(cli, "positional")
    assert r.returncode == 1
    assert r.stderr == "\n[ERROR] First input token (positional) is not a registered flag; parser does not support positional arguments!" + tail
    r = run(cli, "-a", "-s", "3", "-s", "4")
    assert r.returncode == 1 and r.stderr == "\n[ERROR] Duplicate flag -s!" + tail
    r = run(cli, "-a", "-s", "x")
    assert r.returncode == 1 and 'Conversion failed for string "x"!' in r.stderr
    r = run(cli, "-a", "-o", "onlyprefix")
    assert r.returncode == 1 and "Not enough arguments for flag -o!" in r.stderr


def test_help_exits_zero(cli):
    r = run(cli, "-h")
    assert r.returncode == 0 and "-auto-priors" in r.stdout
    r = run(cli, "--help")
    assert r.returncode == 0


@pytest.mark.parametrize("case", ["c1_fb", "k4_mixed_scheme", "mv_c22"])
def test_segments_rule_of_the_driver_on_the_reference_binarys_own_sequences(case, tmp_path):
    """`-O segments` (reference src/Records.hpp:208-209: #marginal segments and the length of StateMarginals' count queue,
    src/StateMarginals.hpp:51-137,204, written BEFORE the sweep's last run is added): the driver's rule
    (hammlet::MarginalSegmentSets, include/hammlet/Records.hpp) fed with the recorded sweeps of a `sequences` file the unmodified
    reference binary wrote must print the `segments` file the same run wrote (tests/golden/<case>/) - host logic, no GPU."""
    from hammlet_amd import build
    build.build_library()
    exe = str(tmp_path / "segment_sets")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(REPO, "include"), "-o", exe,
                    os.path.join(REPO, "tests", "fixtures", "segment_sets_driver.cpp"), "-L", os.path.join(REPO, "hammlet_amd"), "-lhammlet_hip",
                    "-Wl,-rpath," + os.path.join(REPO, "hammlet_amd")], check=True)
    r = subprocess.run([exe, os.path.join(GOLD, case, "sequences.csv")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(GOLD, case, "segments.csv")).read()
