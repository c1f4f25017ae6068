"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker / reported baseline.  The product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB) or (
        os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "rtc_oracle.cpp"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class SynthDesc(C.Structure):
    _fields_ = [("fam_seed", C.c_uint64), ("mut_seed", C.c_uint64),
                ("mut_thr", C.c_uint32), ("n_every", C.c_uint32)]


class KssdParams(C.Structure):
    _fields_ = [("half_k", C.c_int), ("half_subk", C.c_int), ("drlevel", C.c_int),
                ("kmer_size", C.c_int), ("use64", C.c_int), ("dim_size", C.c_int),
                ("dim_end", C.c_int), ("id", C.c_int)]


class Edge(C.Structure):
    _fields_ = [("pre", C.c_int), ("suf", C.c_int), ("dist", C.c_double)]


class CEdge(C.Structure):
    _fields_ = [("pre", C.c_int), ("suf", C.c_int), ("common", C.c_uint32)]


class TuneResult(C.Structure):
    _fields_ = [("kmer_size", C.c_int), ("contain_compress", C.c_int),
                ("is_containment", C.c_int), ("ok", C.c_int), ("max_dist", C.c_double)]


EDGE_DT = np.dtype([("pre", "<i4"), ("suf", "<i4"), ("dist", "<f8")])
CEDGE_DT = np.dtype([("pre", "<i4"), ("suf", "<i4"), ("common", "<u4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_murmur3_smhasher_verification.restype = C.c_uint32
        L.orc_mh_kmer_hash.restype = C.c_uint64
        L.orc_mh_kmer_hash.argtypes = [C.c_char_p, C.c_int, C.c_uint32]
        L.orc_kssd_shuffle_dim.restype = C.POINTER(C.c_int)
        L.orc_kssd_sketch.restype = C.c_uint64
        L.orc_common_u64.restype = C.c_uint32
        L.orc_common_u32.restype = C.c_uint32
        L.orc_mst_distance.restype = C.c_double
        L.orc_mst_distance.argtypes = [C.c_int] * 5
        L.orc_mst_radio.restype = C.c_int
        L.orc_mst_radio.argtypes = [C.c_double, C.c_int]
        L.orc_greedy_distance.restype = C.c_double
        L.orc_greedy_distance.argtypes = [C.c_int] * 5
        L.orc_kssd_greedy_distance.restype = C.c_double
        L.orc_kssd_greedy_distance.argtypes = [C.c_int] * 4
        L.orc_candidate_pairs.restype = C.c_uint64
        L.orc_mst.restype = C.c_uint64
        L.orc_kruskal.restype = C.c_uint64
        L.orc_forest_clusters.restype = C.c_uint32
        L.orc_greedy_minhash.restype = C.c_uint32
        L.orc_greedy_kssd.restype = C.c_uint32
        L.orc_tune_parameters.restype = TuneResult
        L.orc_tune_parameters.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                          C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def minhash_impl():
    """What the batch sketcher's inner loop is (reported with bench.py's cpu_baseline)."""
    lanes = lib().orc_minhash_impl_avx2()
    return {8: "8-lane AVX-512 (vpmullq) MurmurHash3 port", 4: "4-lane AVX2 MurmurHash3 port"}.get(lanes, "scalar MurmurHash3 port")


def murmur3_x64_128(data: bytes, seed: int):
    out = (C.c_uint64 * 2)()
    lib().orc_murmur3_x64_128(data, C.c_int(len(data)), C.c_uint32(seed), out)
    return int(out[0]), int(out[1])


def kmer_hash(kmer: bytes, seed=42):
    return int(lib().orc_mh_kmer_hash(kmer, len(kmer), seed))


def synth_genome(fam_seed, mut_seed, mut_thr, length, n_every=0, pos0=0):
    d = SynthDesc(fam_seed, mut_seed, mut_thr, n_every)
    out = np.empty(length, dtype=np.uint8)
    lib().orc_synth_genome(C.byref(d), C.c_uint64(pos0), C.c_uint64(length), _p(out))
    return out


def sketch_minhash_batch(seq: np.ndarray, off: np.ndarray, k: int, sizes, seed=42, threads=0):
    """seq u8 concatenated genomes, off u64[n+1].  Returns list of ascending u64 arrays."""
    n = len(off) - 1
    sizes = np.ascontiguousarray(np.broadcast_to(np.asarray(sizes, dtype=np.uint32), (n,)))
    stride = int(sizes.max()) if n else 0
    out = np.zeros((n, max(stride, 1)), dtype=np.uint64)
    cnt = np.zeros(n, dtype=np.uint32)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    lib().orc_sketch_minhash_batch(_p(seq), _p(off), C.c_uint32(n), C.c_int(k), C.c_uint32(seed),
                                   _p(sizes), _p(out), C.c_uint32(max(stride, 1)), _p(cnt),
                                   C.c_int(threads))
    return [out[i, :cnt[i]].copy() for i in range(n)]


def kssd_params(kmer_size, drlevel):
    p = KssdParams()
    lib().orc_kssd_params_init(kmer_size, drlevel, C.byref(p))
    return p


_shuffle_cache = {}


def kssd_shuffle_dim(half_subk):
    if half_subk not in _shuffle_cache:
        ptr = lib().orc_kssd_shuffle_dim(half_subk)
        n = 1 << (4 * half_subk)
        _shuffle_cache[half_subk] = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
    return _shuffle_cache[half_subk]


def kssd_sketch(seq: np.ndarray, kmer_size=21, drlevel=3):
    p = kssd_params(kmer_size, drlevel)
    sd = kssd_shuffle_dim(p.half_subk)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    cap = max(1024, len(seq))
    o32 = np.zeros(cap, dtype=np.uint32)
    o64 = np.zeros(cap if p.use64 else 1, dtype=np.uint64)
    n = lib().orc_kssd_sketch(C.byref(p), _p(sd), _p(seq), C.c_uint64(len(seq)), _p(o32), _p(o64),
                              C.c_uint64(cap))
    return (o64[:n].copy() if p.use64 else o32[:n].copy())


def sketch_kssd_batch(seq, off, shuffled_dim, kmer_size=21, drlevel=3, threads=1):
    """KSSD sketches of concatenated genomes (one orc_kssd_sketch call per genome; the calls release
    the GIL, so a thread pool spreads them over `threads` cores like the reference's OpenMP loop over
    files, src/SketchInfo.cpp:1067)."""
    from concurrent.futures import ThreadPoolExecutor
    p = kssd_params(kmer_size, drlevel)
    sd = np.ascontiguousarray(shuffled_dim, dtype=np.int32)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    n = len(off) - 1
    L = lib()

    def one(g):
        a, b = int(off[g]), int(off[g + 1])
        cap = (b - a) // (16 ** drlevel) * 2 + 4096
        while True:
            o32 = np.empty(cap, dtype=np.uint32)
            o64 = np.empty(cap if p.use64 else 1, dtype=np.uint64)
            m = L.orc_kssd_sketch(C.byref(p), _p(sd), C.c_void_p(seq.ctypes.data + a), C.c_uint64(b - a), _p(o32), _p(o64),
                                  C.c_uint64(cap))
            if m <= cap:
                return (o64[:m].copy() if p.use64 else o32[:m].copy())
            cap = int(m)

    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        return list(ex.map(one, range(n)))


def to_csr(sketches, dtype=np.uint64):
    lens = np.array([len(s) for s in sketches], dtype=np.uint32)
    start = np.zeros(len(sketches), dtype=np.uint64)
    if len(sketches) > 1:
        start[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    flat = (np.concatenate([np.asarray(s, dtype=dtype) for s in sketches])
            if len(sketches) and lens.sum() else np.zeros(0, dtype=dtype))
    return np.ascontiguousarray(flat, dtype=dtype), start, lens


def common(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype == np.uint32:
        return int(lib().orc_common_u32(_p(a), C.c_uint32(len(a)), _p(b), C.c_uint32(len(b))))
    a = a.astype(np.uint64, copy=False); b = b.astype(np.uint64, copy=False)
    return int(lib().orc_common_u64(_p(a), C.c_uint32(len(a)), _p(b), C.c_uint32(len(b))))


def candidate_pairs(flat, start, lens):
    n = len(lens)
    width = flat.dtype.itemsize
    cap = max(1, n * (n - 1) // 2)
    out = np.zeros(cap, dtype=CEDGE_DT)
    m = lib().orc_candidate_pairs(_p(flat), C.c_int(width), _p(start), _p(lens), C.c_uint32(n),
                                  _p(out), C.c_uint64(cap))
    return out[:m].copy()


def mst(flat, start, lens, kmer_size, is_containment, threshold, threads=1):
    n = len(lens)
    out = np.zeros(max(n, 1), dtype=EDGE_DT)
    m = lib().orc_mst(_p(flat), C.c_int(flat.dtype.itemsize), _p(start), _p(lens), C.c_uint32(n),
                      C.c_int(kmer_size), C.c_int(int(is_containment)), C.c_double(threshold),
                      C.c_int(threads), _p(out))
    return out[:m].copy()


def forest_clusters(mst_edges, threshold, n):
    mst_edges = np.ascontiguousarray(mst_edges, dtype=EDGE_DT)
    order = np.zeros(max(n, 1), dtype=np.int32)
    off = np.zeros(n + 1, dtype=np.uint32)
    nc = lib().orc_forest_clusters(_p(mst_edges), C.c_uint64(len(mst_edges)), C.c_double(threshold),
                                   C.c_int(n), _p(order), _p(off))
    return [order[off[i]:off[i + 1]].tolist() for i in range(nc)]


def greedy_minhash(flat, start, lens, size_cfg, kmer_size, is_containment, threshold):
    n = len(lens)
    rep = np.zeros(max(n, 1), dtype=np.int32)
    size_cfg = np.ascontiguousarray(np.broadcast_to(np.asarray(size_cfg, dtype=np.uint32), (n,)))
    nc = lib().orc_greedy_minhash(_p(flat), _p(start), _p(lens), _p(size_cfg), C.c_uint32(n),
                                  C.c_int(kmer_size), C.c_int(int(is_containment)),
                                  C.c_double(threshold), _p(rep))
    return int(nc), rep[:n].copy()


def greedy_kssd(flat, start, lens, kmer_size, threshold):
    n = len(lens)
    rep = np.zeros(max(n, 1), dtype=np.int32)
    nc = lib().orc_greedy_kssd(_p(flat), C.c_int(flat.dtype.itemsize), _p(start), _p(lens),
                               C.c_uint32(n), C.c_int(kmer_size), C.c_double(threshold), _p(rep))
    return int(nc), rep[:n].copy()


def tune_parameters(greedy, is_set_kmer, is_containment, is_jaccard, kmer_size, threshold,
                    contain_compress, sketch_size, max_size, min_size, avg_size):
    return lib().orc_tune_parameters(int(greedy), int(is_set_kmer), int(is_containment),
                                     int(is_jaccard), kmer_size, threshold, contain_compress,
                                     sketch_size, max_size, min_size, avg_size)
