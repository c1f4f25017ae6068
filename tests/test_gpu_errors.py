"""GPU: error behaviour of the C ABI -- bad arguments are rejected with a status and a message,
unsupported requests fail loudly, nothing falls back silently."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_argument_validation(ctx):
    from rabbittclust_amd import _lib, api
    lib = ctx.lib
    seq = torch.zeros(4096, dtype=torch.uint8, device=ctx.device)
    out = torch.zeros(1000, dtype=torch.int64, device=ctx.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    off = np.array([0, 1000], dtype=np.uint64)
    p = lambda t: C.c_void_p(t.data_ptr())
    np_p = lambda a: a.ctypes.data_as(C.c_void_p)
    # k out of range
    for k in (0, 33, -1):
        st = lib.rtc_sketch_minhash_dev(ctx.h, p(seq), np_p(off), 1, k, 42, None, 1000, p(out), 1000, p(cnt))
        assert st == _lib.RTC_ERR_ARG and b"k=" in lib.rtc_last_error(ctx.h)
    # misaligned sequence pointer
    st = lib.rtc_sketch_minhash_dev(ctx.h, C.c_void_p(seq.data_ptr() + 1), np_p(off), 1, 21, 42, None, 1000, p(out), 1000, p(cnt))
    assert st == _lib.RTC_ERR_ARG and b"aligned" in lib.rtc_last_error(ctx.h)
    # sketch size larger than the output stride; non-monotone offsets; null pointers
    assert lib.rtc_sketch_minhash_dev(ctx.h, p(seq), np_p(off), 1, 21, 42, None, 2000, p(out), 1000, p(cnt)) == _lib.RTC_ERR_ARG
    bad = np.array([100, 0], dtype=np.uint64)
    assert lib.rtc_sketch_minhash_dev(ctx.h, p(seq), np_p(bad), 1, 21, 42, None, 1000, p(out), 1000, p(cnt)) == _lib.RTC_ERR_ARG
    assert lib.rtc_sketch_minhash_dev(ctx.h, None, np_p(off), 1, 21, 42, None, 1000, p(out), 1000, p(cnt)) == _lib.RTC_ERR_ARG
    assert lib.rtc_sketch_minhash_dev(None, p(seq), np_p(off), 1, 21, 42, None, 1000, p(out), 1000, p(cnt)) == _lib.RTC_ERR_ARG
    # n == 0 is a no-op
    assert lib.rtc_sketch_minhash_dev(ctx.h, p(seq), np_p(off), 0, 21, 42, None, 1000, p(out), 1000, p(cnt)) == _lib.RTC_OK
    # pair kernel: bad width, tile outside [0,n), unknown algo
    sk = api.SketchSet.from_host([np.arange(5, dtype=np.uint64)] * 3, ctx.device)
    com = torch.zeros((3, 3), dtype=torch.int32, device=ctx.device)
    args = lambda width, r1, algo: (ctx.h, p(sk.hashes), width, p(sk.start), p(sk.len), 3, 0, r1, 0, 3, p(com), 3, 0, algo)
    assert lib.rtc_pair_common_dev(*args(2, 3, 0)) == _lib.RTC_ERR_ARG
    assert lib.rtc_pair_common_dev(*args(8, 4, 0)) == _lib.RTC_ERR_ARG
    assert lib.rtc_pair_common_dev(*args(8, 3, 7)) == _lib.RTC_ERR_ARG
    assert lib.rtc_pair_common_dev(*args(8, 3, 0)) == _lib.RTC_OK
    # KSSD: kmer_size / drlevel ranges and the table-size limit (the reference overflows an int there)
    sd = np.zeros(16, dtype=np.int32)
    w, need = C.c_int(), C.c_uint32()
    kss = lambda k, dr: lib.rtc_sketch_kssd_dev(ctx.h, p(seq), np_p(off), 1, k, dr, np_p(sd), p(out), 100, p(cnt), C.byref(w), C.byref(need))
    assert kss(40, 3) == _lib.RTC_ERR_ARG
    assert kss(21, 9) == _lib.RTC_ERR_ARG
    assert kss(31, 6) == _lib.RTC_ERR_UNSUPPORTED
    # greedy without the configured sizes
    rep = np.zeros(3, dtype=np.int32); ncl = C.c_uint32()
    st = lib.rtc_greedy(ctx.h, p(sk.hashes), 8, p(sk.start), p(sk.len), 3, None, 21, 0, 0, 0.05, np_p(rep), C.byref(ncl))
    assert st == _lib.RTC_ERR_ARG


def test_empty_and_degenerate_inputs(ctx, oracle):
    from rabbittclust_amd import api
    # all-empty sketches: every count is zero, MST is empty, greedy makes everyone a representative
    sk = api.SketchSet.from_host([np.zeros(0, dtype=np.uint64)] * 5, ctx.device)
    assert int(ctx.pair_common(sk, algo=0).abs().sum()) == 0
    assert len(ctx.mst(sk, 0.05)) == 0
    n, rep = ctx.greedy(sk, 0.05, size_cfg=1000)
    assert n == 5 and rep.tolist() == [0, 1, 2, 3, 4]
    # a single genome; two identical genomes
    one = api.SketchSet.from_host([np.arange(10, dtype=np.uint64)], ctx.device)
    assert len(ctx.mst(one, 0.05)) == 0
    two = api.SketchSet.from_host([np.arange(10, dtype=np.uint64)] * 2, ctx.device)
    m = ctx.mst(two, 0.05)
    assert len(m) == 1 and m[0]["dist"] == 0.0 and (int(m[0]["preNode"]), int(m[0]["sufNode"])) == (1, 0)
