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
