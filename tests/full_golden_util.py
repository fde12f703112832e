"""Full-size golden runs of the unmodified reference binary (tests/golden/full/, made by tests/golden/make_full_golden.py):
manifest, trace regeneration and file access shared by the CPU and the GPU test."""
import hashlib
import json
import lzma
import os

from tests import oracle_lib as ol

FULL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full")
with open(os.path.join(FULL, "manifest.json")) as f:
    MANIFEST = json.load(f)


def matches_golden(case, output, path):
    """does the file at `path` hold the bytes the reference binary wrote?  (a file above 256 MB would be kept as its checksum)"""
    e = MANIFEST[case]["files"][output]
    if e["file"] is not None:
        with open(path, "rb") as f:
            return f.read() == golden_bytes(case, output)
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for piece in iter(lambda: f.read(1 << 24), b""):
            h.update(piece)
    return os.path.getsize(path) == e["bytes"] and h.hexdigest() == e["sha256"]


def golden_bytes(case, output):
    """the reference's file as it was written (stored xz-compressed above 1 MB)"""
    e = MANIFEST[case]["files"][output]
    path = os.path.join(FULL, case, e["file"])
    if path.endswith(".xz"):
        with lzma.open(path, "rb") as f:
            data = f.read()
    else:
        with open(path, "rb") as f:
            data = f.read()
    assert hashlib.sha256(data).hexdigest() == e["sha256"], "golden file damaged: %s/%s" % (case, output)
    return data


def trace(case):
    """the case's float32 trace from the repository's generator; must be the one the reference binary was fed"""
    m = MANIFEST[case]
    x = ol.trace(m["T"], m["trace_levels"], m["data_seed"])
    assert hashlib.sha256(x.tobytes()).hexdigest() == m["trace_sha256"], "the generator no longer yields the golden run's trace"
    return x


def enough_memory(case, per_position_bytes):
    """full-size cases are skipped on boxes that cannot hold them (a size guard, not a result)"""
    try:
        import psutil
        return psutil.virtual_memory().available > MANIFEST[case]["T"] * per_position_bytes
    except ImportError:
        return True
