"""Development build of libhammlet_hip.so for ONE number of states (seconds instead of minutes): the same sources with
-DHML_ONLY_K=<K>, written to hammlet_amd/libhammlet_hip_k<K>.so.  Use with HML_LIBRARY=<that file> for A/B timing
(hammlet_amd.capi.load_library honours the variable).
usage: python tools/dev_build.py 5 [extra hipcc flags...]"""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from hammlet_amd import build as b  # noqa: E402

K = int(sys.argv[1])
tag = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
out = os.path.join(b.PKG_DIR, "libhammlet_hip_k%d%s.so" % (K, tag))
t0 = time.time()
units = sorted(f for f in os.listdir(b.CSRC) if f.endswith(".hip"))
objs = []
os.makedirs(os.path.join(b.CSRC, "build", "dev"), exist_ok=True)
procs = []
for u in units:
    obj = os.path.join(b.CSRC, "build", "dev", u[:-4] + ".k%d%s.o" % (K, tag))
    objs.append(obj)
    procs.append(subprocess.Popen([b._hipcc()] + b.HIPCC_FLAGS + ["-DHML_ONLY_K=%d" % K] + extra + ["-c", "-o", obj, os.path.join(b.CSRC, u)],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
for p in procs:
    _, err = p.communicate()
    if p.returncode:
        sys.exit("\n".join(l for l in err.splitlines() if "error" in l or "note:" in l))
subprocess.run([b._hipcc()] + b.HIPCC_FLAGS + ["-shared", "-o", out] + objs, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
print(out, "%.1f s" % (time.time() - t0))
