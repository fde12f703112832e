"""Randomised differential run: GPU (through the C ABI) against the CPU checker in device mode on random configurations -
sizes, states, data dimensions, schemes, priors, weight multipliers, forward geometry.  Not part of the test suite (it
takes minutes); prints one line per configuration and stops at the first difference.
    python tools/fuzz_parity.py [n_configs=100] [seed=1] [many|compat|wide]   (many: several chains through hml_iterate_many; compat: the
    reference-compatible mode against the checker's reference mode; wide: the path for more than 16 states, 2-64 states)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hammlet_amd as hml

from tests.fuzz_util import fuzz

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fuzz(hml, n_cfg, int(sys.argv[2]) if len(sys.argv) > 2 else 1, log=lambda s: print(s, flush=True), many=len(sys.argv) > 3 and sys.argv[3] == "many", compat=len(sys.argv) > 3 and sys.argv[3] == "compat",
     wide=len(sys.argv) > 3 and sys.argv[3] == "wide")
print("all %d configurations identical" % n_cfg)
