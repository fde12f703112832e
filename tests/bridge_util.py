"""Posterior summaries used by the statistical bridge tests (GPU chain on Philox streams vs the reference's output
files): per-position state probabilities relabelled by ascending posterior-mean emission mean, the arg-max
segmentation, and posterior-mean parameters."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAIN_MODE_TOL = 0.05   # a chain is in the main posterior mode when its sorted posterior-mean means are this close to the data levels


def manifest():
    with open(os.path.join(GOLDEN, "bridge_manifest.json")) as f:
        return json.load(f)


def parse_marginals(text, K):
    """marginals file (reference src/StateMarginals.hpp:282-305) -> (segment lengths [M], counts [M][K])"""
    seg, cnt = [], []
    for line in text.strip().split("\n"):
        f = line.split("\t")
        seg.append(int(f[0]))
        c = [int(v) for v in f[1:]]
        cnt.append(c + [0] * (K - len(c)))      # trailing states that never occurred are not printed
    return np.asarray(seg, np.int64), np.asarray(cnt, np.int64)


def parse_parameters(text, K):
    """parameters file: one line per recorded sweep, mean \t variance per state -> [n][K][2]"""
    rows = [[float(v) for v in line.split("\t")] for line in text.strip().split("\n")]
    return np.asarray(rows, np.float64).reshape(len(rows), K, 2)


def summarise(seg, cnt, params):
    """-> dict: order (labels by ascending posterior-mean mean), prob [T][K] relabelled, argmax [T], mean/var [K]"""
    pm = params.mean(axis=0)                      # [K][2]
    order = np.argsort(pm[:, 0], kind="stable")
    n = cnt.sum(axis=1)
    assert np.all(n == n[0])
    prob_seg = cnt[:, order] / float(n[0])
    prob = np.repeat(prob_seg, seg, axis=0)
    return {"order": order, "prob": prob, "argmax": prob.argmax(axis=1), "mean": pm[order, 0], "var": pm[order, 1], "recorded": int(n[0])}


def distance(a, b):
    """(fraction of positions whose arg-max state differs, mean total-variation distance of the per-position state
    distributions, largest |difference| of the posterior-mean means, largest relative difference of the variances)"""
    return (float((a["argmax"] != b["argmax"]).mean()), float(0.5 * np.abs(a["prob"] - b["prob"]).sum(axis=1).mean()),
            float(np.abs(a["mean"] - b["mean"]).max()), float((np.abs(a["var"] - b["var"]) / b["var"]).max()))


def reference_summary(name, seed, K):
    d = os.path.join(GOLDEN, name)
    with open(os.path.join(d, "marginals_seed%d.csv" % seed)) as f:
        seg, cnt = parse_marginals(f.read(), K)
    with open(os.path.join(d, "parameters_seed%d.csv" % seed)) as f:
        par = parse_parameters(f.read(), K)
    return summarise(seg, cnt, par)


def in_main_mode(summary, case):
    return bool(np.abs(summary["mean"] - np.asarray(case["levels"], float)).max() < MAIN_MODE_TOL)


def yardstick(case, full_refs):
    """Tolerances from the reference's own seed-to-seed spread: arg-max disagreement and total-variation distance
    between the two reference runs whose files are committed, largest pairwise difference of the posterior-mean
    parameters among ALL reference runs in the main mode."""
    runs = [r for r in case["reference_runs"] if r["main_mode"]]
    m = np.asarray([r["mean"] for r in runs])
    v = np.asarray([r["var"] for r in runs])
    d_mean = max(float(np.abs(m[i] - m[j]).max()) for i in range(len(runs)) for j in range(i))
    d_var = max(float((np.abs(v[i] - v[j]) / v[j]).max()) for i in range(len(runs)) for j in range(i))
    d_arg, d_tv, _, _ = distance(full_refs[0], full_refs[1])
    return {"argmax": d_arg, "tv": d_tv, "mean": d_mean, "var": d_var}


def assert_within(dist, yard, what=""):
    """a chain may differ from a reference run by twice what two reference runs differ by (plus one count in the
    last recorded digit of the summaries)"""
    d_arg, d_tv, d_mean, d_var = dist
    assert d_arg <= 2 * yard["argmax"] + 1e-4, (what, "arg-max segmentation", dist, yard)
    assert d_tv <= 2 * yard["tv"] + 1e-4, (what, "state marginals", dist, yard)
    assert d_mean <= 2 * yard["mean"], (what, "posterior-mean means", dist, yard)
    assert d_var <= 2 * yard["var"], (what, "posterior-mean variances", dist, yard)


# ---- run-length forms for full-size traces (10^8 positions: a dense [T][K] table per chain would be 4 GB) -------------------------
def summarise_rle(seg, cnt, params):
    """like summarise, but the per-position quantities stay per marginal segment: `ends` (cumulative segment ends), `prob_seg`
    [M][K] relabelled by ascending posterior-mean mean, `argmax_seg` [M]"""
    pm = params.mean(axis=0)
    order = np.argsort(pm[:, 0], kind="stable")
    n = cnt.sum(axis=1)
    assert np.all(n == n[0])
    prob_seg = cnt[:, order] / float(n[0])
    return {"order": order, "ends": np.cumsum(seg), "prob_seg": prob_seg, "argmax_seg": prob_seg.argmax(axis=1),
            "mean": pm[order, 0], "var": pm[order, 1], "recorded": int(n[0])}


def distance_rle(a, b):
    """distance() for two run-length summaries: both segmentations are refined to their common one and every piece weighs
    in with its length"""
    assert a["ends"][-1] == b["ends"][-1]
    ends = np.union1d(a["ends"], b["ends"])
    length = np.diff(np.concatenate(([0], ends))).astype(np.float64)
    ia = np.searchsorted(a["ends"], ends, side="left")
    ib = np.searchsorted(b["ends"], ends, side="left")
    T = float(ends[-1])
    d_arg = float((length * (a["argmax_seg"][ia] != b["argmax_seg"][ib])).sum() / T)
    d_tv = float((length * 0.5 * np.abs(a["prob_seg"][ia] - b["prob_seg"][ib]).sum(axis=1)).sum() / T)
    return (d_arg, d_tv, float(np.abs(a["mean"] - b["mean"]).max()), float((np.abs(a["var"] - b["var"]) / b["var"]).max()))


def read_golden_text(path):
    """a golden text file, stored plain or xz-compressed (files above 1 MB)"""
    import lzma
    if os.path.exists(path):
        with open(path) as f:
            return f.read()
    with lzma.open(path + ".xz", "rt") as f:
        return f.read()


def reference_summary_rle(name, seed, K):
    d = os.path.join(GOLDEN, name)
    seg, cnt = parse_marginals(read_golden_text(os.path.join(d, "marginals_seed%d.csv" % seed)), K)
    par = parse_parameters(read_golden_text(os.path.join(d, "parameters_seed%d.csv" % seed)), K)
    return summarise_rle(seg, cnt, par)


def yardstick_rle(case, full_refs):
    runs = [r for r in case["reference_runs"] if r["main_mode"]]
    m = np.asarray([r["mean"] for r in runs])
    v = np.asarray([r["var"] for r in runs])
    d_mean = max(float(np.abs(m[i] - m[j]).max()) for i in range(len(runs)) for j in range(i))
    d_var = max(float((np.abs(v[i] - v[j]) / v[j]).max()) for i in range(len(runs)) for j in range(i))
    d_arg, d_tv, _, _ = distance_rle(full_refs[0], full_refs[1])
    return {"argmax": d_arg, "tv": d_tv, "mean": d_mean, "var": d_var}
