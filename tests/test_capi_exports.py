"""The C-ABI library loads without a GPU and exports every symbol include/hml.h declares."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "hml.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(hml_[a-z0-9_]+)\s*\(", text))
    names.discard("hml_record_cb")
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from hammlet_amd import build, capi
    build.build_library()
    lib = ctypes.CDLL(build.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), s
    # the Python mirror binds exactly the declared surface
    assert sorted(capi.SIGNATURES) == syms


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import hammlet_amd
    with pytest.raises(hammlet_amd.HmlError):
        hammlet_amd.Chain()


def test_product_never_references_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for root in ("hammlet_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(REPO, root)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, fn)).read().lower()
                    assert "oracle" not in txt, os.path.join(dp, fn)
