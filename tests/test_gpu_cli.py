"""End-to-end file-level parity of the `hammlet` driver (GPU) with the CPU checker's driver in device mode:
same flags, same input, byte-identical marginals / sequences / blocks / parameters / compression / segments files."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "hammlet_amd", "hammlet")
ORACLE_CLI = os.path.join(ol.ORACLE_DIR, "hammlet_oracle")
OUTS = ["marginals", "sequences", "parameters", "blocks", "compression", "segments"]


def run_pair(x, flags, text_input=False):
    ol.load()
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "in.f32")
        x.tofile(raw)
        if text_input:
            txt = os.path.join(tmp, "in.txt")
            np.savetxt(txt, x, fmt="%.9g")
            g_in = ["-f", txt]
        else:
            g_in = ["-raw", raw]
        g = subprocess.run([CLI] + g_in + ["-o", os.path.join(tmp, "g-"), ".csv", "-a"] + flags + ["-O"] + OUTS,
                           capture_output=True, text=True)
        o = subprocess.run([ORACLE_CLI, "--raw", raw, "-o", os.path.join(tmp, "o-"), ".csv", "-a"] + flags + ["-O"] + OUTS +
                           ["--rng", "2", "--math", "1", "--reduce", "1"], capture_output=True, text=True)
        assert g.returncode == 0, g.stderr
        assert o.returncode == 0, o.stderr
        res = {}
        for name in OUTS:
            res[name] = (open(os.path.join(tmp, "g-%s.csv" % name)).read(), open(os.path.join(tmp, "o-%s.csv" % name)).read())
        return res, g.stdout, o.stdout


@pytest.mark.parametrize("T,K,flags,text", [
    (100000, 3, "-s 3 -R 1 -i F 100 1", True),                                  # BASELINE config 1
    (100000, 3, "-s 3 -R 11", False),                                            # default scheme M 500 0 S P F 200 0 F 300 3
    (20000, 4, "-s 4 -R 3 -i M 50 5 D F 60 2 P M 10 1 S F 30 1", False),
    (100000, 3, "-s 3 -R 5 -S -i F 50 1", False),
    (100000, 3, "-s 3 -R 7 -m 1.5 -t 1 10 -I 2 -e normal 0.1 0.8 -i M 20 1 F 50 1", False),
    (100000, 3, "-s 6 -R 8 -t 0.1 -i S F 50 1 P D F 20 1", False),
    (300000, 5, "-s 5 -R 2 -i F 30 3", False),
    (60000, 6, "-s 20 -R 12 -i F 30 1", False),                                  # more than 16 states: the default path takes them (round 5)
    (30000, 5, "-s 40 -R 13 -t 0.2 2 -i M 10 1 S P F 15 2 D F 5 1", True),
])
def test_cli_files_equal_checker_files(T, K, flags, text):
    x = ol.trace(T, K, 1)
    res, g_out, o_out = run_pair(x, flags.split(), text_input=text)
    for name, (g, o) in res.items():
        assert g == o, name
    assert sorted(g_out.splitlines()) == sorted(o_out.splitlines())


def test_cli_errors_like_the_reference():
    x = ol.trace(1000, 3, 1)
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "in.f32")
        x.tofile(raw)
        r = subprocess.run([CLI, "-raw", raw, "-o", os.path.join(tmp, "g-"), ".csv"], capture_output=True, text=True)
        assert r.returncode == 1
        assert r.stderr == "\n[ERROR] Manual theta priors not implemented, use -a!\nTerminating HaMMLET. The rest is silence.\n"
        r = subprocess.run([CLI, "-raw", raw, "-o", os.path.join(tmp, "h-"), ".csv", "-a", "-i", "F", "10"], capture_output=True, text=True)
        assert r.returncode == 1 and "must be multiples of 3" in r.stderr
        ok = subprocess.run([CLI, "-raw", raw, "-o", os.path.join(tmp, "g-"), ".csv", "-a", "-i", "F", "5", "1"], capture_output=True, text=True)
        assert ok.returncode == 0
        again = subprocess.run([CLI, "-raw", raw, "-o", os.path.join(tmp, "g-"), ".csv", "-a", "-i", "F", "5", "1"], capture_output=True, text=True)
        assert again.returncode == 1 and "already exists! Use -w to allow overwrite!" in again.stderr
        r = subprocess.run([CLI, "-raw", raw, "-a", "-a"], capture_output=True, text=True)
        assert r.returncode == 1 and "Duplicate flag -a!" in r.stderr


@pytest.mark.parametrize("P,D,T,flags", [(2, 2, 30001, "-R 4 -i F 30 1"), (3, 2, 20000, "-R 5 -i M 10 1 S P F 20 2"), (2, 3, 65537, "-R 6 -S -i F 15 1")])
def test_cli_multivariate_files_equal_checker_files(P, D, T, flags):
    """`-s C P D` through the driver: interleaved text input, P parameter pairs per line of the parameters file"""
    x = np.stack([ol.trace(T, min(P, 5), 70 + d) for d in range(D)], axis=1).reshape(-1)
    res, g_out, o_out = run_pair(x, ("-s C %d %d " % (P, D) + flags).split(), text_input=True)
    for name, (g, o) in res.items():
        assert g == o, name
    assert len(res["parameters"][0].splitlines()[0].split("\t")) == 2 * P


def test_cli_chains_pools_marginals_over_rccl(tmp_path):
    """`hammlet -chains 3` (extension): three independent chains (Philox sub-keys 0, 1, 2) in their own host threads, the
    recorded marginals pooled by hml_allreduce_marginals (RCCL) before PREFIXmarginalsSUFFIX is written.  Expected result
    from three single-chain runs of the same driver: relabel each by ascending last mean, sum the dense counts, cut at
    the union of the boundaries."""
    from hammlet_amd import chains
    T, K = 50000, 3
    x = ol.trace(T, K, 6)
    raw = str(tmp_path / "in.f32")
    x.tofile(raw)
    common = ["-raw", raw, "-a", "-s", str(K), "-R", "5", "-i", "F", "40", "4", "-w"]
    r = subprocess.run([CLI] + common + ["-chains", "3", "-o", str(tmp_path / "pool-"), ".csv", "-O", "marginals", "parameters", "maxsegmentation"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dense = np.zeros((K, T), np.int64)
    bnd = np.zeros(T, np.int64)
    for k in range(3):
        pre = str(tmp_path / ("one%d-" % k))
        s = subprocess.run([CLI] + common + ["-chain", str(k), "-o", pre, ".csv", "-O", "marginals", "parameters"], capture_output=True, text=True)
        assert s.returncode == 0, s.stderr
        # the pooled run's per-chain side files are those of the single-chain runs
        side = "pool-parameters.csv" if k == 0 else "pool-chain%d.parameters.csv" % k
        assert open(str(tmp_path / side)).read() == open(pre + "parameters.csv").read()
        rows = [list(map(int, l.split("\t"))) for l in open(pre + "marginals.csv").read().strip().split("\n")]
        last = [float(v) for v in open(pre + "parameters.csv").read().strip().split("\n")[-1].split("\t")]
        perm = chains.relabel_permutation(np.array(last[0::2], np.float32))
        # the pooled files use common labels; PREFIX[chainK.]relabelSUFFIX holds the chain's own label of pooled state 0, 1, ...
        rel = "pool-relabel.csv" if k == 0 else "pool-chain%d.relabel.csv" % k
        assert open(str(tmp_path / rel)).read() == "\t".join(str(int(v)) for v in perm) + "\n"
        seg = np.array([r_[0] for r_ in rows])
        cnt = np.array([r_[1:] + [0] * (K + 1 - len(r_)) for r_ in rows])
        starts = np.concatenate([[0], np.cumsum(seg)[:-1]])
        dense += np.repeat(cnt, seg, axis=0).T[perm]
        bnd[starts] = 1
    starts = np.flatnonzero(bnd)
    seg = np.diff(np.append(starts, T))
    cols = int(np.flatnonzero(dense.any(axis=1)).max()) + 1
    want = "".join("%d\t%s\n" % (n, "\t".join(str(v) for v in dense[:cols, s])) for n, s in zip(seg, starts))
    got = open(str(tmp_path / "pool-marginals.csv")).read()
    assert got == want
    assert all(sum(map(int, l.split("\t")[1:])) == 30 for l in got.strip().split("\n"))
    assert open(str(tmp_path / "pool-maxsegmentation.csv")).read() == ol.max_segmentation_text(got)
    # nothing to pool when neither marginals nor their segmentation are asked for: no relabel files, no RCCL
    r = subprocess.run([CLI] + common + ["-chains", "2", "-o", str(tmp_path / "np-"), ".csv", "-O", "parameters"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert not (tmp_path / "np-relabel.csv").exists() and (tmp_path / "np-chain1.parameters.csv").exists()
