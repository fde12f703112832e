"""Statistical bridge GPU -> REFERENCE FILES.  The GPU chain draws from Philox streams (design deviation D1), the
reference from one sequential mt19937, so equal seeds are different chains and no file can be compared byte for byte.
What must agree is the posterior: the GPU chains' marginals and posterior-mean parameters are compared with output
files of the unmodified reference binary (tests/golden/bridge_*, made by tests/golden/make_bridge_golden.py), and the
yardstick for every tolerance is the spread among reference runs that differ only in their seed.
(reference src/StateMarginals.hpp:268-310 writes the marginals, src/Records.hpp:196-203 the parameters.)"""
import numpy as np
import pytest

from tests import bridge_util as bu

pytestmark = pytest.mark.gpu


def gpu_summary(hml, x, K, scheme, seed):
    """one GPU chain through the C ABI: marginals + the parameters of every recorded sweep"""
    g = hml.Chain(device=0, seed=seed)
    g.load(x)
    g.set_model(K, g.autoprior(0.2, 0.9))
    g.sample_prior()
    rows = []
    g.set_recording(marginals=True, callback=lambda ch, i: rows.append(ch.theta().astype(np.float64)))
    toks = scheme.split()[1:]
    for i in range(0, len(toks), 3):
        g.iterate(toks[i], int(toks[i + 1]), int(toks[i + 2]))
    g.sync()
    seg, cnt = g.marginals_rle()
    cnt = np.pad(cnt, ((0, 0), (0, K - cnt.shape[1])))
    par = np.asarray(rows).reshape(len(rows), K, 2)
    g.close()
    return bu.summarise(np.asarray(seg, np.int64), np.asarray(cnt, np.int64), par)


@pytest.mark.parametrize("name,n_chains", [("bridge_c1", 6), ("bridge_k5", 32)])
def test_gpu_posterior_matches_reference_files(hml, name, n_chains):
    c = bu.manifest()[name]
    K = int(c["flags"].split()[1])
    x = hml.synth_gauss(c["T"], len(c["levels"]), c["levels"], c["sigma"], c["dwell"], c["data_seed"])
    ref = [bu.reference_summary(name, s, K) for s in c["seeds"]]
    yard = bu.yardstick(c, ref)
    chains = [gpu_summary(hml, x, K, c["scheme"], seed) for seed in range(1, n_chains + 1)]
    main = [g for g in chains if bu.in_main_mode(g, c)]
    # how often a chain reaches the main mode: same rate as the reference's (binomial, 3 sigma of the difference)
    p_ref = np.mean([r["main_mode"] for r in c["reference_runs"]])
    n_ref = len(c["reference_runs"])
    p_gpu = len(main) / n_chains
    sd = np.sqrt(max(p_ref * (1 - p_ref), 0.0) * (1.0 / n_chains + 1.0 / n_ref))
    assert abs(p_gpu - p_ref) <= 3 * sd + 1e-9, (p_gpu, p_ref, sd)
    assert main, "no GPU chain reached the reference's posterior mode"
    for g in main:
        for r in ref:
            bu.assert_within(bu.distance(g, r), yard, name)


def gpu_summary_rle(hml, x, K, scheme, seed):
    """gpu_summary with the per-position quantities kept per marginal segment (10^8 positions)"""
    g = hml.Chain(device=0, seed=seed)
    g.load(x)
    g.set_model(K, g.autoprior(0.2, 0.9))
    g.sample_prior()
    rows = []
    g.set_recording(marginals=True, callback=lambda ch, i: rows.append(ch.theta().astype(np.float64)))
    toks = scheme.split()[1:]
    for i in range(0, len(toks), 3):
        g.iterate(toks[i], int(toks[i + 1]), int(toks[i + 2]))
    g.sync()
    seg, cnt = g.marginals_rle()
    cnt = np.pad(cnt, ((0, 0), (0, K - cnt.shape[1])))
    par = np.asarray(rows).reshape(len(rows), K, 2)
    g.close()
    return bu.summarise_rle(np.asarray(seg, np.int64), np.asarray(cnt, np.int64), par)


def test_gpu_posterior_matches_reference_files_at_full_size(hml):
    """The bridge at BASELINE config 3's FULL size (round 5): 24 chains of the DEFAULT path on the
    10^8-position trace of the headline workload against marginals and parameters files the unmodified reference binary wrote
    for that trace (tests/golden/bridge_c3, tests/golden/make_bridge_full_golden.py: six seeds, 100 burn-in sweeps, 100 sweeps
    of which every 10th is recorded) - the same yardstick rules as above, the distances taken segment by segment.  Within 200
    sweeps 3 of the reference's 6 runs and 6 of these 24 chains reach the main mode (the others sit in the same two local modes -
    a level split in two, sorted means (-2, -1, -0.55, 0, 1.51) and (-1.5, 0, 0.66, 1, 2) - in both implementations:
    profiles/round5_bridge_c3_probe.txt), hence 24 chains."""
    name = "bridge_c3"
    c = bu.manifest().get(name)
    if c is None:
        pytest.skip("tests/golden/bridge_c3 not generated")
    K = int(c["flags"].split()[1])
    x = hml.synth_gauss(c["T"], len(c["levels"]), c["levels"], c["sigma"], c["dwell"], c["data_seed"], nthreads=16)
    ref = [bu.reference_summary_rle(name, s, K) for s in c["seeds"]]
    yard = bu.yardstick_rle(c, ref)
    n_chains = 24
    chains = [gpu_summary_rle(hml, x, K, c["scheme"], seed) for seed in range(1, n_chains + 1)]
    main = [g for g in chains if bu.in_main_mode(g, c)]
    p_ref = np.mean([r["main_mode"] for r in c["reference_runs"]])
    n_ref = len(c["reference_runs"])
    p_gpu = len(main) / n_chains
    sd = np.sqrt(max(p_ref * (1 - p_ref), 0.0) * (1.0 / n_chains + 1.0 / n_ref))
    assert abs(p_gpu - p_ref) <= 3 * sd + 1e-9, (p_gpu, p_ref, sd)
    assert main, "no GPU chain reached the reference's posterior mode"
    assert all(g["recorded"] == ref[0]["recorded"] for g in main)
    for g in main:
        for r in ref:
            bu.assert_within(bu.distance_rle(g, r), yard, name)


def _posterior_signal(seg, cnt, params):
    """the posterior-mean signal of a run: per marginal segment sum_s p(s) mean_s (mean_s: average of the state's recorded means) - a
    summary that does not depend on how the states are labelled or on states that stay empty (models with more states than levels)"""
    n = cnt.sum(axis=1)
    mean_s = params.mean(axis=0)[:, 0]
    return np.repeat((cnt * mean_s[None, :]).sum(axis=1) / n, seg)


def _run_scheme(g, tokens):
    pending = True
    for tok in tokens:
        if pending:
            g.sample_prior()
            pending = False
        if tok == "P":
            pending = True
        elif tok == "S":
            g.set_static_blocks()
        elif tok == "D":
            g.set_dynamic(True)
        else:
            g.iterate(*tok)
    g.sync()


@pytest.mark.parametrize("case", ["k20_many_states", "k40_mixed_scheme"])
def test_many_state_models_against_reference_files(hml, case):
    """More than 16 states on the DEFAULT path (round 5, hml_k_wide.h) against the files the unmodified reference binary wrote for
    the same input and scheme (tests/golden/k20_many_states, k40_mixed_scheme).  Equal seeds are different chains (D1), and with
    more states than levels the labelling is arbitrary, so runs are compared through the posterior-mean signal
    sum_s p_t(s) mean_s and its error against the true piecewise-constant signal.  These short schemes (30 sweeps recorded from
    the first on) end in one of a few modes - errors of about 0.05, 0.2 or 0.55 at 20 states - in the reference as well, so one
    reference run is no yardstick by itself.  The yardstick is the reference's OWN chain at other seeds: the
    reference-compatible mode, which at the golden run's seed reproduces the golden files (checked here: its signal equals the
    file's) - twelve chains of each kind must show the same share of runs as good as the reference's, the same median error,
    and the good default-path runs must lie as close to the reference file as the good reference-compatible ones do.
    Reference: src/main.cpp:112-137 (any -s K), src/StateMarginals.hpp:268-310, src/Records.hpp:196-203."""
    import json
    import os
    from tests import oracle_lib as ol
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    m = json.load(open(os.path.join(gold, "manifest.json")))[case]
    fl = m["flags"].split()
    K = int(fl[fl.index("-s") + 1])
    golden_seed = int(fl[fl.index("-R") + 1])
    t_off, t_diag = 0.5, 0.5
    if "-t" in fl:
        t_off, t_diag = float(fl[fl.index("-t") + 1]), float(fl[fl.index("-t") + 2])
    toks, sch = [], fl[fl.index("-i") + 1:]
    i = 0
    while i < len(sch):
        if sch[i] in ("P", "S", "D"):
            toks.append(sch[i]); i += 1
        else:
            toks.append((sch[i], int(sch[i + 1]), int(sch[i + 2]))); i += 3
    L = m["trace_levels"]
    x, truth_states = hml.synth_gauss(m["T"], L, ol.LEVELS[L], ol.SIGMA[L], ol.DWELL[L], m["data_seed"], with_states=True)
    assert np.array_equal(x, ol.trace(m["T"], L, m["data_seed"]))
    truth = np.asarray(ol.LEVELS[L], np.float64)[truth_states]
    seg, cnt = bu.parse_marginals(open(os.path.join(gold, case, "marginals.csv")).read(), K)
    ref = _posterior_signal(seg, cnt, bu.parse_parameters(open(os.path.join(gold, case, "parameters.csv")).read(), K))
    rmse = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))

    def run(seed, compat):
        g = hml.Chain(device=0, seed=seed)
        if compat:
            g.set_option("compat", 1)
        g.load(x)
        g.set_model(K, g.autoprior(0.2, 0.9), t_off, t_diag)
        rows = []
        g.set_recording(marginals=True, callback=lambda ch, i: rows.append(ch.theta().astype(np.float64)))
        _run_scheme(g, toks)
        gs, gc = g.marginals_rle()
        gc = np.pad(gc, ((0, 0), (0, K - gc.shape[1])))
        sig = _posterior_signal(np.asarray(gs, np.int64), np.asarray(gc, np.int64), np.asarray(rows).reshape(len(rows), K, 2))
        g.close()
        return sig

    # the anchor: the reference's chain at the golden seed IS the golden run (its parameters file prints six decimals)
    assert rmse(run(golden_seed, True), ref) < 1e-5
    n = 12
    seeds = [s for s in range(1, n + 2) if s != golden_seed][:n]
    err_ref = rmse(ref, truth)
    good_enough = 1.5 * err_ref + 0.01
    stats = {}
    for kind in ("default", "compat"):
        sigs = [run(s, kind == "compat") for s in seeds]
        errs = np.asarray([rmse(s, truth) for s in sigs])
        good = errs <= good_enough
        stats[kind] = {"errs": errs, "good": float(good.mean()), "median": float(np.median(errs)),
                       "to_ref": [rmse(s, ref) for s, ok in zip(sigs, good) if ok]}
    d, c = stats["default"], stats["compat"]
    assert abs(d["good"] - c["good"]) <= 0.5, (case, d["good"], c["good"])          # (3 sigma of two binomial shares of 12)
    # (the errors are bimodal: compare the share of runs that are clearly off as well, not a median)
    off_d, off_c = float((d["errs"] > 5 * err_ref).mean()), float((c["errs"] > 5 * err_ref).mean())
    assert abs(off_d - off_c) <= 0.5, (case, off_d, off_c)
    assert d["errs"].max() <= 1.5 * c["errs"].max() + 0.05, (case, d["errs"].max(), c["errs"].max())
    if d["to_ref"] and c["to_ref"]:
        assert max(d["to_ref"]) <= 2.0 * max(c["to_ref"]) + 0.02, (case, d["to_ref"], c["to_ref"])
    assert d["errs"].min() <= 1.2 * c["errs"].min() + 0.02, (case, d["errs"].min(), c["errs"].min())
