import sys, numpy as np
sys.path.insert(0, '.')
import hammlet_amd as hml
from tests import oracle_lib as ol
rng = np.random.default_rng(5)
n = 1 << 20
a = rng.choice(np.array([0.5, 0.1, 0.9, 1.0, 1.5, 2.0, 2.5, 17.5, 1000.5, 123456.5], np.float32), n)
b = rng.choice(np.array([1.0, 0.37, 12.5], np.float32), n)
dev = hml.debug_eval(5, a, b, seed=77); host = ol.debug_eval(5, a, b, seed=77)
bad = np.flatnonzero(dev.view(np.uint32) != host.view(np.uint32))
print(len(bad))
for fn in (12,13,14,15,16,17,18,19,20,21):
    d = hml.debug_eval(fn, a, b, seed=77); h = ol.debug_eval(fn, a, b, seed=77)
    bb = np.flatnonzero(d.view(np.uint32) != h.view(np.uint32))
    print("fn", fn, "mismatches", len(bb), "of which in gamma-bad", np.isin(bb, bad).sum())
    for i in bb[:6]: print("   ", i, a[i], d[i], h[i])
    if fn in (15, 20):
        print("   decisions at gamma-bad (dev):", np.bincount(d[bad].astype(int), minlength=4), "host:", np.bincount(h[bad].astype(int), minlength=4))
vb3d = hml.debug_eval(18, a, b, seed=77); dec2 = hml.debug_eval(20, a, b, seed=77)
malpha = np.where(a < 1, a + 1, a).astype(np.float32)
a1 = (malpha - np.float32(1.0)/np.float32(3.0)).astype(np.float32)
exp2 = ((a1 * vb3d).astype(np.float32) * b).astype(np.float32)
sel = bad[(dec2[bad] != 3) & (a[bad] >= 1)]
print("second-iteration accepts with alpha>=1:", len(sel))
print(" host == expected:", (host[sel].view(np.uint32) == exp2[sel].view(np.uint32)).sum(), " dev == expected:", (dev[sel].view(np.uint32) == exp2[sel].view(np.uint32)).sum())
v3d = hml.debug_eval(13, a, b, seed=77)
exp1 = ((a1 * v3d).astype(np.float32) * b).astype(np.float32)
print(" host == first-iter value:", (host[sel].view(np.uint32) == exp1[sel].view(np.uint32)).sum(), " dev == first-iter value:", (dev[sel].view(np.uint32) == exp1[sel].view(np.uint32)).sum())
nw = hml.debug_eval(21, a, b, seed=77).astype(int)
dec1 = hml.debug_eval(15, a, b, seed=77)
rej = np.flatnonzero(dec1 == 3)
print("all first-iteration rejections:", len(rej), "words-consumed hist:", np.bincount(nw[rej], minlength=12)[:14])
print("bad cases words-consumed hist:", np.bincount(nw[bad], minlength=12)[:14])
good_rej = np.setdiff1d(rej, bad)
print("good rejections hist:", np.bincount(nw[good_rej], minlength=12)[:14])
# neighbours in the same wave: lane index and whether neighbours rejected
print("bad lane idx hist (i%64):", np.bincount(bad % 64, minlength=64))
n1 = hml.debug_eval(12, a, b, seed=77); n2v = hml.debug_eval(16, a, b, seed=77); a2v = hml.debug_eval(17, a, b, seed=77)
print("i alpha  n1  n2  n_dev_implied  n_host_implied")
for i in sel[:12]:
    nd_ = (np.cbrt(dev[i] / (a1[i] * b[i])) - 1) / a2v[i]
    nh_ = (np.cbrt(host[i] / (a1[i] * b[i])) - 1) / a2v[i]
    print(i, a[i], n1[i], n2v[i], nd_, nh_)

n3 = hml.debug_eval(22, a, b, seed=77); n3s = hml.debug_eval(23, a, b, seed=77)
print("i n_dev_implied n3 n3saved")
for i in sel[:12]:
    nd_ = (np.cbrt(dev[i] / (a1[i] * b[i])) - 1) / a2v[i]
    print(i, nd_, n3[i], n3s[i])

dev1 = hml.debug_eval(24, a, b, seed=77)
print("single-lane gamma mismatches vs host:", (dev1.view(np.uint32) != host.view(np.uint32)).sum())
