"""ctypes loader for librtclust_host.so: the C++ host side of the drop-in (FASTA reading, parameter
tuning, on-disk formats, KSSD shuffle table) as a small C ABI (rabbittclust_amd/host/host_capi.cpp).
No GPU code; used by bench.py (--mode kssd needs generate_shuffle_dim) and the tests."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtclust_host.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(LIB_PATH)
        _lib.rtch_shuffle_dim.restype = C.c_int
        _lib.rtch_shuffle_dim.argtypes = [C.c_int, C.c_void_p]
    return _lib


_shuffle_cache = {}


def generate_shuffle_dim(half_subk):
    """generate_shuffle_dim (src/SketchInfo.cpp:60-102): the glibc srand/rand shuffle table, int32[2^(4*half_subk)]."""
    if half_subk not in _shuffle_cache:
        sd = np.zeros(1 << (4 * half_subk), dtype=np.int32)
        n = load().rtch_shuffle_dim(int(half_subk), sd.ctypes.data_as(C.c_void_p))
        assert n == len(sd)
        _shuffle_cache[half_subk] = sd
    return _shuffle_cache[half_subk]
