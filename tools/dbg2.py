import sys, numpy as np
sys.path.insert(0, '.')
import hammlet_amd as hml
from tests import oracle_lib as ol
T, K = 200000, 5
x = ol.trace(T, K, 7)
o = ol.OracleChain(K=K, seed=42, rng=ol.RNG_CTR, math=ol.MATH_DEV, reduce=ol.REDUCE_DEV); o.load(x)
g = hml.Chain(device=0, seed=42); g.load(x)
po = o.autoprior(); pg = g.autoprior()
o.init_model(); g.set_model(K, pg)
o.token("F"); g.sample_prior()
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
for it in range(25):
    o.iterate("F", 1, 0); g.iterate("F", 1, 0); g.sync()
    ok_b = np.array_equal(o.blocks(), g.blocks())
    ok_q = ok_b and np.array_equal(o.states(), g.states())
    ok_t = np.array_equal(bits(o.theta()), bits(g.theta()))
    Ao, po_ = o.transitions(); Ag, pg_ = g.transitions()
    ok_A = np.array_equal(bits(Ao), bits(Ag)); ok_p = np.array_equal(bits(po_), bits(pg_))
    st = g.stats()
    print(it, "B", o.num_blocks(), g.num_blocks(), "blocks", ok_b, "states", ok_q, "theta", ok_t, "A", ok_A, "pi", ok_p, st)
    if not (ok_b and ok_q and ok_t and ok_A and ok_p):
        if ok_b and not ok_q:
            d = np.flatnonzero(o.states() != g.states()); print("first state diffs", d[:10], len(d))
        co = o.counts(); cg = g.counts()
        print("trans eq", np.array_equal(co[0], cg[0]), "occ eq", np.array_equal(co[1], cg[1]), "sum eq", np.array_equal(bits(co[2]), bits(cg[2])), np.array_equal(bits(co[3]), bits(cg[3])))
        print(co[0]); print(cg[0]); print(co[2], cg[2])
        break
