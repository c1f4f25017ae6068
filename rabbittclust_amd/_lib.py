"""ctypes loader for librtclust_hip.so (the C ABI declared in include/rtclust.h).

The HIP extension IS the product: there is no CPU fallback.  Importing this module without the
built library raises; calling any op without a GPU returns RTC_ERR_HIP which `check` turns into
an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTC_HIP_LIB") or os.path.join(_HERE, "librtclust_hip.so")  # override: alternative builds

RTC_OK, RTC_ERR_ARG, RTC_ERR_HIP, RTC_ERR_UNSUPPORTED, RTC_ERR_OVERFLOW, RTC_ERR_NOMEM, RTC_ERR_COMM = range(7)
STATUS_NAMES = {0: "RTC_OK", 1: "RTC_ERR_ARG", 2: "RTC_ERR_HIP", 3: "RTC_ERR_UNSUPPORTED",
                4: "RTC_ERR_OVERFLOW", 5: "RTC_ERR_NOMEM", 6: "RTC_ERR_COMM"}


class RtcError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class SynthDesc(C.Structure):
    _fields_ = [("fam_seed", C.c_uint64), ("mut_seed", C.c_uint64),
                ("mut_thr", C.c_uint32), ("n_every", C.c_uint32)]


class CEdge(C.Structure):
    _fields_ = [("i", C.c_uint32), ("j", C.c_uint32), ("common", C.c_uint32)]


class ShardStats(C.Structure):
    _fields_ = [("row0", C.c_uint32), ("row1", C.c_uint32), ("cand_edges", C.c_uint64), ("rounds", C.c_uint32),
                ("s_fixed", C.c_uint32), ("contractions", C.c_uint32), ("pad", C.c_uint32),
                ("pair_ms", C.c_float), ("mst_ms", C.c_float)]


class Edge(C.Structure):
    _fields_ = [("preNode", C.c_int32), ("sufNode", C.c_int32), ("dist", C.c_double)]


# every symbol include/rtclust.h declares, with its ctypes signature
_vp, _u32, _u64, _i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
SIGNATURES = {
    "rtc_device_count": (_i, []),
    "rtc_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "rtc_ctx_destroy": (None, [_vp]),
    "rtc_warmup": (_i, [_i]),
    "rtc_ctx_reload_options": (_i, [_vp]),
    "rtc_ctx_set_stream": (_i, [_vp, _vp]),
    "rtc_ctx_own_stream": (_i, [_vp]),
    "rtc_ctx_sync": (_i, [_vp]),
    "rtc_last_error": (C.c_char_p, [_vp]),
    "rtc_version": (C.c_char_p, []),
    "rtc_device_info": (_i, [_vp, C.POINTER(_i)]),
    "rtc_dev_alloc": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "rtc_dev_free": (_i, [_vp, _vp]),
    "rtc_dev_mem_info": (_i, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rtc_copy_h2d": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "rtc_copy_d2h": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "rtc_memset_dev": (_i, [_vp, _vp, _i, C.c_size_t]),
    "rtc_host_alloc": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "rtc_host_free": (_i, [_vp, _vp]),
    "rtc_unpack_bases_dev": (_i, [_vp, _vp, _u64, _vp, _u64, _vp]),
    "rtc_timer_start": (_i, [_vp]),
    "rtc_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "rtc_synth_genomes_dev": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "rtc_sketch_minhash_dev": (_i, [_vp, _vp, _vp, _u32, _i, _u32, _vp, _u32, _vp, _u32, _vp]),
    "rtc_sketch_minhash_packed_dev": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u32, _i, _u32, _vp, _u32, _vp, _u32, _vp]),
    "rtc_sketch_kssd_dev": (_i, [_vp, _vp, _vp, _u32, _i, _i, _vp, _vp, _u32, _vp,
                                 C.POINTER(_i), C.POINTER(_u32)]),
    "rtc_sketch_kssd_packed_dev": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u32, _i, _i, _vp, _vp, _u32, _vp,
                                        C.POINTER(_i), C.POINTER(_u32)]),
    "rtc_pair_common_dev": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _u64,
                                 _i, _i]),
    "rtc_pair_mash_dev": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u64]),
    "rtc_extract_edges_dev": (_i, [_vp, _vp, _u64, _u32, _u32, _u32, _u32, _vp, _i, _vp, _u64, _vp]),
    "rtc_pair_last_path": (_i, [_vp]),
    "rtc_diag_counters": (_i, [_vp, _vp]),
    "rtc_pair_last_kernel_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "rtc_pair_edges_dev": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _i, _vp, _u64, _vp]),
    "rtc_boruvka_key_bits": (_i, [_u32, _u32]),
    "rtc_boruvka_minkey_dev": (_i, [_vp, _vp, _u64, _vp, _u32, _u32, _vp]),
    "rtc_boruvka_init_dev": (_i, [_vp, _u32, _vp, _vp]),
    "rtc_boruvka_union_dev": (_i, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_u32)]),
    "rtc_boruvka_minweight_dev": (_i, [_vp, _vp, _u64, _vp, _i, _vp, _u32, _vp]),
    "rtc_boruvka_minedge_dev": (_i, [_vp, _vp, _u64, _vp, _i, _vp, _u32, _vp, _vp]),
    "rtc_boruvka_fetch_dev": (_i, [_vp, _vp, _u64, _vp, _u32, _vp, _vp]),
    "rtc_msf_dev": (_i, [_vp, _vp, _u64, _vp, _u32, _i, _vp, C.POINTER(_u64), C.POINTER(_i)]),
    "rtc_boruvka_merge_host": (_i, [_u32, _vp, _vp, _vp, _vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rtc_edges_to_mst_host": (_i, [_vp, _u64, _vp, _i, _i, _vp]),
    "rtc_comm_unique_id": (_i, [_vp]),
    "rtc_comm_init_rank": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "rtc_comm_init_all": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp)]),
    "rtc_comm_destroy": (None, [_vp]),
    "rtc_comm_rank": (_i, [_vp]),
    "rtc_comm_size": (_i, [_vp]),
    "rtc_comm_backend": (C.c_char_p, [_vp]),
    "rtc_comm_all_reduce": (_i, [_vp, _vp, C.c_size_t, _i, _i]),
    "rtc_comm_all_reduce_host": (_i, [_vp, _vp, C.c_size_t, _i]),
    "rtc_comm_gather_rows": (_i, [_vp, _vp, C.c_size_t, _u32, _u32, _u32, _i]),
    "rtc_comm_wait": (_i, [_vp]),
    "rtc_comm_broadcast": (_i, [_vp, _vp, C.c_size_t, _i]),
    "rtc_triangle_rows": (_i, [_u32, _i, C.c_double, _vp]),
    "rtc_sketch_minhash_sharded": (_i, [_vp, _vp, _vp, _vp, _u32, _i, _u32, _vp, _u32, _vp, _u32, _vp]),
    "rtc_sketch_minhash_packed_sharded": (_i, [_vp, _vp, _vp, _u64, _vp, _u64, _vp, _u32, _u32, _u32, _i, _i, _u32, _vp, _u32, _vp,
                                               _u32, _vp]),
    "rtc_sketch_kssd_packed_sharded": (_i, [_vp, _vp, _vp, _u64, _vp, _u64, _vp, _u32, _u32, _u32, _i, _i, _i, _vp, _vp, _u32, _vp,
                                            C.POINTER(_i), C.POINTER(_u32)]),
    "rtc_mst_sharded": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _u32, _i, _i, C.c_double, _vp, C.POINTER(_u64), _vp]),
    "rtc_mst": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _i, _i, C.c_double, _vp, C.POINTER(_u64)]),
    "rtc_mst_append": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _u32, _i, _i, C.c_double, _vp, C.POINTER(_u64)]),
    "rtc_mst_dense": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _u32, _i, _i, C.c_double, _vp, C.POINTER(_u64), _i, _vp, _vp]),
    "rtc_greedy_mash": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _i, _i, _u32, C.c_double, _vp, C.POINTER(_u32)]),
    "rtc_mst_mash": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _u32, _i, _i, _u32, _vp, C.POINTER(_u64), _i, _vp, _vp]),
    "rtc_greedy": (_i, [_vp, _vp, _i, _vp, _vp, _u32, _vp, _i, _i, _i, C.c_double, _vp,
                        C.POINTER(_u32)]),
}

_lib = None


def load():
    """Loads the shared library (no GPU needed to load; needed to run anything)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  It must be the first HIP
        # runtime in the process so that our DT_NEEDED entry binds to it; two runtimes in one
        # process cannot both see the GPU.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
