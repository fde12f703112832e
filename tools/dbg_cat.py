import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hammlet_amd as hml
rng = np.random.default_rng(1)
for fn, K in ((30, 16), (31, 5)):
    n = 200000
    kind = rng.integers(0, 4, (n, K))
    w = np.where(kind == 0, 0.0, np.where(kind == 1, rng.integers(0, 10**6, (n, K)) * 2.0 ** -rng.integers(0, 120, (n, K)), rng.integers(0, 1000, (n, K)) / 1000.0)).astype(np.float32)
    u = rng.random((n, 1)).astype(np.float32) * np.ones((1, K), np.float32)
    out = hml.debug_eval(fn, w.ravel(), u.ravel().astype(np.float32)).reshape(n, K)
    comb, lit, uns = out[:, 0], out[:, 1], out[:, 2]
    # numpy literal
    sd = w.astype(np.float64)
    S = np.zeros(n)
    for i in range(K): S = S + sd[:, i]
    cp = np.zeros(n); res = np.full(n, K - 1); done = np.zeros(n, bool)
    with np.errstate(all="ignore"):
        for i in range(K):
            cp = cp + sd[:, i] / S
            c = np.ones(n) if i == K - 1 else cp
            hit = ~done & ~(c < u[:, 0].astype(np.float64))
            res[hit] = i; done |= hit
    print("K", K, "gpu combined != gpu literal:", int((comb != lit).sum()), " gpu literal != numpy:", int((lit != res).sum()), "unsure", int(uns.sum()))
    bad = np.nonzero(comb != lit)[0][:3]
    for j in bad: print(w[j], u[j, 0], comb[j], lit[j], uns[j])
