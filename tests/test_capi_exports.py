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


def test_library_holds_the_sweep_of_every_number_of_states():
    """hammlet_amd/build.py links sixteen objects: the C ABI and, per number of states 2 ... 16, the sweep behind its table of
    function pointers (csrc/hml_capi.hip once, csrc/hml_sweep.hip with -DHML_TU_K=k) - one table per K must be there, and the objects of the
    build must be the ones the layout names (a stale single-object build would define the C ABI twice or not at all)."""
    from hammlet_amd import build
    build.build_library()
    objs = [o for _, o, _ in build._objects()]
    assert "hml_capi.o" in objs and all("hml_sweep_k%d.o" % k in objs for k in range(2, 17))
    lib = ctypes.CDLL(build.LIB_PATH)
    for k in range(2, 17):
        tab = (ctypes.c_void_p * 5).in_dll(lib, "hml_ktab_%d" % k)
        assert all(tab[i] for i in range(5)), k       # sweep, iterate_many, params, compat_draw, derive
    with pytest.raises(ValueError):
        ctypes.c_void_p.in_dll(lib, "hml_ktab_17")


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
