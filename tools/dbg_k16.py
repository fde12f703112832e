import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hammlet_amd as hml
from tests import oracle_lib as ol
from tests.test_gpu_parity import make_pair, setup_model, bits
T, K = 30000, 16
x, o, g = make_pair(hml, T, K, 7, 42)
setup_model(o, g, K)
o.token("F"); g.sample_prior()
o.set_probes(True); g.enable_probes(True)
for it in range(6):
    o.iterate("F", 1, 0); g.iterate("F", 1, 0); g.sync()
    sb = np.array_equal(o.blocks(), g.blocks())
    so, sg = o.states(), g.states()
    rows_same = np.array_equal(bits(o.forward_rows()), bits(g.forward_rows()))
    nd = np.nonzero(so != sg)[0] if so.size == sg.size else None
    print("sweep", it, "blocks", sb, "rows", rows_same, "B", so.size, "state diffs", None if nd is None else (nd.size, nd[:10], so[nd[:10]], sg[nd[:10]]), g.stats()["forward_refits"], g.stats()["forward_serial"])
    if nd is not None and nd.size:
        break
