"""CPU suite: the C-ABI library loads and exports every symbol include/rtclust.h declares; the
product path refuses to run without a GPU (no CPU fallback); nothing in the product imports oracle/."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rtclust.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from rabbittclust_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rtclust.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes SIGNATURES out of sync with include/rtclust.h"
    assert b"gfx950" in lib.rtc_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rabbittclust_amd import api, _lib
    with pytest.raises(_lib.RtcError):
        api.Context(0)
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rtc_ctx_create(0, C.byref(h)) != 0  # fails loudly, no silent host path


def test_product_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "rabbittclust_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"\boracle\b", src) and "pyoracle" in src or "liboracle" in src or "rtc_oracle" in src:
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_host_merge_helper_unions_components():
    """rtc_boruvka_merge_host is pure host code: exercise it without a GPU."""
    import numpy as np
    from rabbittclust_amd import _lib
    from rabbittclust_amd.api import CEDGE_DT
    lib = _lib.load()
    n = 6
    NONE = 0x7FFFFFFFFFFFFFFF
    # components {0},{1},{2},{3},{4},{5}; winners: 0-1 (mutual), 2->1, 4-5 (mutual), 3 none
    ekey = np.array([(1 << 32) | 0, (1 << 32) | 0, (2 << 32) | 1, NONE, (5 << 32) | 4, (5 << 32) | 4], dtype=np.uint64)
    ecommon = np.array([7, 7, 3, 0, 9, 9], dtype=np.uint32)
    comp = np.arange(n, dtype=np.uint32)
    sel = np.zeros(n, dtype=CEDGE_DT)
    nsel, added = C.c_uint64(0), C.c_uint64(0)
    st = lib.rtc_boruvka_merge_host(n, ekey.ctypes.data_as(C.c_void_p), ecommon.ctypes.data_as(C.c_void_p),
                                    comp.ctypes.data_as(C.c_void_p), sel.ctypes.data_as(C.c_void_p),
                                    C.byref(nsel), C.byref(added))
    assert st == 0 and added.value == 3 and nsel.value == 3
    assert comp.tolist() == [0, 0, 0, 3, 4, 4]
    assert sorted((int(e["i"]), int(e["j"]), int(e["common"])) for e in sel[:3]) == [(1, 0, 7), (2, 1, 3), (5, 4, 9)]
