"""The statistical bridge of tests/test_gpu_reference_bridge.py with the CPU checker in device mode in the GPU's place
(the GPU equals the checker in that mode bit for bit): Philox-addressed draws, hml_math transcendental functions, tree
sums and exact integer counts against output files of the unmodified reference binary, yardstick = the spread among
reference runs that differ only in their seed."""
import numpy as np
import pytest

from tests import bridge_util as bu
from tests import oracle_lib as ol


def checker_summary(x, K, scheme, seed):
    o = ol.OracleChain(K=K, seed=seed, rng=ol.RNG_CTR, math=ol.MATH_DEV, reduce=ol.REDUCE_DEV)
    o.load(x)
    o.autoprior()
    o.init_model()
    o.set_record(marginals=True, params=True)
    o.token("F")
    toks = scheme.split()[1:]
    for i in range(0, len(toks), 3):
        o.iterate(toks[i], int(toks[i + 1]), int(toks[i + 2]))
    seg, cnt = bu.parse_marginals(o.text("marginals"), K)
    par = bu.parse_parameters(o.text("parameters"), K)
    o.close()
    return bu.summarise(seg, cnt, par)


@pytest.mark.parametrize("name,n_chains", [("bridge_c1", 4), ("bridge_k5", 24)])
def test_device_mode_posterior_matches_reference_files(name, n_chains):
    c = bu.manifest()[name]
    K = int(c["flags"].split()[1])
    x = ol.synth_gauss(c["T"], len(c["levels"]), c["levels"], c["sigma"], c["dwell"], c["data_seed"])
    ref = [bu.reference_summary(name, s, K) for s in c["seeds"]]
    yard = bu.yardstick(c, ref)
    chains = [checker_summary(x, K, c["scheme"], seed) for seed in range(1, n_chains + 1)]
    main = [g for g in chains if bu.in_main_mode(g, c)]
    p_ref = np.mean([r["main_mode"] for r in c["reference_runs"]])
    n_ref = len(c["reference_runs"])
    sd = np.sqrt(max(p_ref * (1 - p_ref), 0.0) * (1.0 / n_chains + 1.0 / n_ref))
    assert abs(len(main) / n_chains - p_ref) <= 3 * sd + 1e-9, (len(main), n_chains, p_ref)
    assert main
    for g in main:
        for r in ref:
            bu.assert_within(bu.distance(g, r), yard, name)
