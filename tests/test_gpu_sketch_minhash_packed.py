"""GPU parity: MinHash sketches straight from the 2-bit staging format (rtc_sketch_minhash_packed_dev) vs the CPU oracle run
over the characters the batch was packed from, AND vs the ASCII kernel (rtc_sketch_minhash_dev) on those characters:
bit-exact hash sets.  Every case family of tests/test_gpu_sketch_minhash.py; the packing is the one the command lines'
parser performs (base codes at 2 bits, everything outside ACGT listed as runs)."""
import os

import numpy as np
import pytest
import torch

from test_gpu_sketch_kssd_packed import pack_batch
from test_gpu_sketch_minhash import _random_genomes

pytestmark = pytest.mark.gpu


def _reload_options():
    """the library reads its RTC_* switches when a context is created: every live context reads them again"""
    from rabbittclust_amd import api
    api.reload_all_options()

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _sketch_packed(ctx, seq, off, k, size=None, sizes=None):
    packed, n_bases, runs = pack_batch(seq)
    d_p = torch.from_numpy(packed).to(ctx.device)
    d_r = torch.from_numpy(runs).to(ctx.device)
    sk = ctx.sketch_minhash_packed(d_p, off, k=k, size=size if size else 1, sizes=sizes, n_bases=n_bases, runs=d_r)
    ctx.sync()
    return sk


def _check(ctx, oracle, seq, off, k, size=None, sizes=None, ascii_too=True):
    got = _sketch_packed(ctx, seq, off, k, size, sizes).to_host()
    want = oracle.sketch_minhash_batch(seq, off, k, sizes if sizes is not None else size)
    assert len(got) == len(want)
    for g, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), f"genome {g}: k={k} got {len(a)} want {len(b)}"
    if ascii_too:
        asc = ctx.sketch_minhash(ctx.upload_sequences(seq), off, k=k, size=size if size else 1, sizes=sizes)
        ctx.sync()
        for g, (a, b) in enumerate(zip(got, asc.to_host())):
            assert np.array_equal(a, b), f"genome {g}: packed and ASCII kernels differ"


@pytest.mark.parametrize("k", [21, 17, 16, 11, 32, 5, 18, 19, 20, 22, 23, 24, 28, 29, 1, 3, 9, 13, 25, 26, 27, 30, 31])
def test_packed_sketch_matches_oracle_various_k(ctx, oracle, k):
    rng = np.random.default_rng(100 + k)
    seq, off = _random_genomes(rng, [50_000, 123_457, 80_001, 15_359, 15_361, 30_720, 262_144 + 77])
    _check(ctx, oracle, seq, off, k, size=1000)


def test_packed_sketch_with_n_runs_and_lowercase(ctx, oracle):
    rng = np.random.default_rng(7)
    seq, off = _random_genomes(rng, [200_000, 100_000, 60_000], n_rate=0.003, lower_rate=0.3)
    _check(ctx, oracle, seq, off, 21, size=1000)
    _check(ctx, oracle, seq, off, 31, size=300)


def test_packed_sketch_short_empty_and_ragged(ctx, oracle):
    rng = np.random.default_rng(8)
    lens = [0, 1, 20, 21, 22, 100, 0, 999, 5000, 16, 15375, 3, 63, 64, 65, 4095, 4096, 4097, 32767, 32768, 32769]
    seq, off = _random_genomes(rng, lens)
    _check(ctx, oracle, seq, off, 21, size=1000)
    _check(ctx, oracle, seq, off, 21, size=50)


def test_packed_sketch_multi_record_separator(ctx, oracle):
    """records of a genome are separated by one character outside ACGT: a run of length one in the staging format"""
    rng = np.random.default_rng(9)
    recs = [rng.choice(ACGT, size=L) for L in (5000, 30, 20, 7000, 21, 40_000, 22, 70_000)]
    sep = np.frombuffer(b"\n", dtype=np.uint8)
    g = np.concatenate([x for r in recs for x in (r, sep)])
    seq = np.concatenate([g, g[::-1].copy()])
    off = np.array([0, len(g), 2 * len(g)], dtype=np.uint64)
    _check(ctx, oracle, seq, off, 21, size=400)


def test_packed_sketch_many_contigs(ctx, oracle):
    """assemblies of hundreds of contigs: waves that meet a separator take the general walk, the others the express walk"""
    rng = np.random.default_rng(19)
    parts, off = [], [0]
    for g in range(6):
        recs = []
        for _ in range(120):
            recs.append(rng.choice(ACGT, size=int(rng.integers(200, 30_000))))
            recs.append(np.frombuffer(b"\n", dtype=np.uint8))
            if rng.random() < 0.1:
                recs.append(np.full(int(rng.integers(1, 200)), ord("N"), dtype=np.uint8))
        s = np.concatenate(recs)
        parts.append(s)
        off.append(off[-1] + len(s))
    _check(ctx, oracle, np.concatenate(parts), np.array(off, dtype=np.uint64), 21, size=1000)


def test_packed_sketch_variable_sizes_containment_mode(ctx, oracle):
    rng = np.random.default_rng(10)
    lens = [400_000, 150_000, 90_000, 1_000_000]
    seq, off = _random_genomes(rng, lens)
    sizes = np.array([max(L // 200, 100) for L in lens], dtype=np.uint32)  # fileBytes/compress
    _check(ctx, oracle, seq, off, 21, sizes=sizes)


def test_packed_sketch_repetitive_genome(ctx, oracle):
    rng = np.random.default_rng(11)
    unit = rng.choice(ACGT, size=3000)
    g = np.tile(unit, 70)
    poly = np.full(50_000, ord("A"), dtype=np.uint8)
    seq = np.concatenate([g, poly])
    off = np.array([0, len(g), len(g) + len(poly)], dtype=np.uint64)
    _check(ctx, oracle, seq, off, 21, size=1000)


def test_packed_sketch_large_genome_is_segmented(ctx, oracle):
    d = oracle.synth_genome(1234, 99, 300, 6_000_000)
    off = np.array([0, len(d)], dtype=np.uint64)
    _check(ctx, oracle, d, off, 21, size=1000)


def test_packed_sketch_size_beyond_one_lds_pass(ctx, oracle):
    """Sketch sizes above 6144 are selected in passes over ascending hash ranges: same bottom-s set."""
    rng = np.random.default_rng(31)
    lens = [3000, 6164, 9000, 12308, 40_000, 700_000, 0, 25]
    seq, off = _random_genomes(rng, lens, n_rate=0.001, lower_rate=0.1)
    _check(ctx, oracle, seq, off, 21, size=8000)
    _check(ctx, oracle, seq, off, 21, size=6145, ascii_too=False)
    _check(ctx, oracle, seq, off, 21, size=12288, ascii_too=False)
    _check(ctx, oracle, seq, off, 17, size=20000)
    sizes = np.array([100, 7000, 20000, 6144, 6145, 13000, 9000, 8000], dtype=np.uint32)
    _check(ctx, oracle, seq, off, 21, sizes=sizes)


def test_packed_sketch_large_size_multi_segment(ctx, oracle):
    rng = np.random.default_rng(32)
    seq, off = _random_genomes(rng, [6_000_000, 50_000], n_rate=0.0005)
    _check(ctx, oracle, seq, off, 21, size=15000)
    rep = np.tile(rng.choice(ACGT, size=7001), 300)
    off2 = np.array([0, len(rep)], dtype=np.uint64)
    _check(ctx, oracle, rep, off2, 21, size=20000)


def test_packed_sketch_argument_errors(ctx):
    from rabbittclust_amd import _lib
    seq = np.frombuffer(b"ACGT" * 1000, dtype=np.uint8)
    packed, n_bases, runs = pack_batch(seq)
    d_p = torch.from_numpy(packed).to(ctx.device)
    d_r = torch.from_numpy(runs).to(ctx.device)
    off = np.array([0, len(seq)], dtype=np.uint64)
    out = torch.empty(1000, dtype=torch.int64, device=ctx.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=ctx.device)

    def call(nb, offs, stride, k=21):
        return ctx.lib.rtc_sketch_minhash_packed_dev(ctx.h, d_p.data_ptr(), nb, d_r.data_ptr(), len(runs) // 2, offs.ctypes.data, 1, k, 42,
                                                     None, 1000, out.data_ptr(), stride, cnt.data_ptr())
    assert call(n_bases, off, 100) == _lib.RTC_ERR_ARG          # stride below the sketch size
    assert call(n_bases - 1, off, 1000) == _lib.RTC_ERR_ARG     # n_bases not a multiple of 64
    assert call(n_bases, np.array([0, n_bases + 64], dtype=np.uint64), 1000) == _lib.RTC_ERR_ARG  # genomes beyond the buffer
    assert call(n_bases, off, 1000, k=33) == _lib.RTC_ERR_ARG
    assert call(n_bases, off, 1000) == _lib.RTC_OK


def test_packed_run_list_contract_is_checked_on_the_device(ctx):
    """runs out of order, overlapping, or beyond the batch: the asynchronous check raises the context's flag, the next
    rtc_ctx_sync (or packed call) returns RTC_ERR_ARG -- once; runs that merely touch are fine"""
    from rabbittclust_amd import _lib
    rng = np.random.default_rng(3)
    seq, off = _random_genomes(rng, [40_000])
    packed, n_bases, _ = pack_batch(seq)
    d_p = torch.from_numpy(packed).to(ctx.device)
    for runs, ok in (([100, 10, 110, 5], True), ([500, 10, 100, 10], False), ([100, 50, 120, 5], False), ([n_bases - 10, 20], False)):
        d_r = torch.tensor(runs, dtype=torch.int64, device=ctx.device)
        ctx.sketch_minhash_packed(d_p, off, k=21, size=100, n_bases=n_bases, runs=d_r)
        if ok:
            ctx.sync()
        else:
            with pytest.raises(_lib.RtcError) as e:
                ctx.sync()
            assert e.value.status == _lib.RTC_ERR_ARG and "run list" in str(e.value)
            ctx.sync()  # reported once


@pytest.mark.parametrize("size", [2035, 2500, 3318, 3319, 3574, 3575, 4000])
def test_packed_sketch_sizes_around_the_third_workgroup_per_cu(ctx, oracle, size):
    """the packed kernel's safe mode appends one k-mer per lane between two looks at the candidate count, so a third
    workgroup per CU fits up to s = 3 318 (3 574 with the packed tables): sizes on both sides of every boundary, on genomes
    that start without a threshold (short ones: everything passes at first, safe mode, merges under the minimum room)"""
    rng = np.random.default_rng(900 + size)
    seq, off = _random_genomes(rng, [300_000, 40_000, 3000, 1_200_000, 70_001], n_rate=0.0005)
    _check(ctx, oracle, seq, off, 21, size=size)


def test_packed_sketch_many_tiny_genomes_and_max_size(ctx, oracle):
    rng = np.random.default_rng(12)
    lens = [int(x) for x in rng.integers(0, 3000, size=600)]
    lens[5] = 0
    seq, off = _random_genomes(rng, lens, n_rate=0.01, lower_rate=0.2)
    _check(ctx, oracle, seq, off, 21, size=1000)
    _check(ctx, oracle, seq, off, 32, size=64)
    seq2, off2 = _random_genomes(rng, [4000, 250_000])
    _check(ctx, oracle, seq2, off2, 21, size=6144)
    _check(ctx, oracle, seq2, off2, 17, size=6144)


def test_packed_starting_threshold_restart_on_low_complexity(ctx, oracle):
    """whole-genome workgroups that must run again from "none" (periodic genomes, a poly-A run) beside normal ones"""
    rng = np.random.default_rng(61)
    parts = []
    for g in range(3400):
        if g % 850 == 0:
            parts.append(np.tile(rng.choice(ACGT, size=3000), 70)[:200_000])
        elif g % 850 == 1:
            parts.append(np.tile(rng.choice(ACGT, size=40), 5000))
        elif g % 850 == 2:
            parts.append(np.full(150_000, ord("A"), dtype=np.uint8))
        else:
            parts.append(rng.choice(ACGT, size=int(rng.integers(40_000, 60_000))))
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    seq = np.concatenate(parts)
    got = _sketch_packed(ctx, seq, off, 21, size=1000).to_host()
    check = [g for g in range(len(parts)) if g % 850 < 4] + [5, 77, 1234, 3399]
    sub_off = np.zeros(len(check) + 1, dtype=np.uint64)
    sub_off[1:] = np.cumsum([len(parts[g]) for g in check])
    want = oracle.sketch_minhash_batch(np.concatenate([parts[g] for g in check]), sub_off, 21, 1000)
    for g, w in zip(check, want):
        assert np.array_equal(got[g], w), g
    assert len(got[0]) == 1000 and len(got[1]) < 100 and len(got[2]) == 1
    asc = ctx.sketch_minhash(ctx.upload_sequences(seq), off, k=21, size=1000)
    ctx.sync()
    for g, (a, b) in enumerate(zip(got, asc.to_host())):
        assert np.array_equal(a, b), g


def test_packed_partial_segments_start_from_the_genome_threshold(ctx, oracle):
    """few large genomes cut into segments that start from the genome's threshold, flagged genomes walked again; other
    threshold factors and segment counts -- identical sketches every way"""
    rng = np.random.default_rng(67)
    parts = [np.tile(rng.choice(ACGT, size=3000), 1000),
             rng.choice(ACGT, size=2_500_000),
             np.tile(rng.choice(ACGT, size=40), 50_000),
             oracle.synth_genome(77, 5, 200, 3_200_000),
             np.full(1_700_000, ord("A"), dtype=np.uint8),
             rng.choice(ACGT, size=1_900_000)]
    parts[3][1_000_000:1_000_050] = ord("N")
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(q) for q in parts])
    seq = np.concatenate(parts)
    want = oracle.sketch_minhash_batch(seq, off, 21, 1000)
    for env in ({}, {"RTC_SKETCH_T0_FACTOR": "1"}, {"RTC_SKETCH_T0_FACTOR": "0"}, {"RTC_SKETCH_ROUNDS": "1"},
                {"RTC_SKETCH_ROUNDS": "4", "RTC_SKETCH_T0_FACTOR": "2"}):
        os.environ.update(env)
        _reload_options()
        try:
            got = _sketch_packed(ctx, seq, off, 21, size=1000).to_host()
        finally:
            for key in env:
                del os.environ[key]
            _reload_options()
        for g, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), (env, g, len(a), len(b))
    assert len(want[0]) == 1000 and len(want[2]) < 100 and len(want[4]) == 1


def test_packed_starting_threshold_factor_does_not_change_results(ctx, oracle):
    """factors that make workgroups restart, disable the threshold, or fill the express walk's per-wave queue (the load is
    handed to the general walk, then safe mode): identical sketches every way, equal to the ASCII kernel's"""
    from rabbittclust_amd import api
    desc = api.synth_family_descs(120, 10, global_seed=9)
    L = 230_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    ref = ctx.sketch_minhash(seq, off, k=21, size=500)
    ctx.sync()
    pb = api.pack_staging(seq, int(off[-1]))
    for f in (None, "1", "0", "8", "40", "2000"):
        if f is not None:
            os.environ["RTC_SKETCH_T0_FACTOR"] = f
            _reload_options()
        try:
            alt = ctx.sketch_minhash_packed(pb, off, k=21, size=500)
            ctx.sync()
        finally:
            if f is not None:
                del os.environ["RTC_SKETCH_T0_FACTOR"]
                _reload_options()
        assert torch.equal(alt.hashes, ref.hashes) and torch.equal(alt.len, ref.len), f


@pytest.mark.parametrize("k", [21, 17, 19, 23, 12, 28, 31, 32, 9])
def test_packed_input_with_the_packed_table_layout(ctx, oracle, k):
    """both LDS table layouts of the hash (RTC_SKETCH_PACKED forces the one with lo(b * c) inside the entries)"""
    rng = np.random.default_rng(300 + k)
    seq, off = _random_genomes(rng, [260_000, 123_457, 15_361, 700_001], n_rate=0.0005, lower_rate=0.01)
    os.environ["RTC_SKETCH_PACKED"] = "1"
    _reload_options()
    try:
        _check(ctx, oracle, seq, off, k, size=1000, ascii_too=False)
    finally:
        del os.environ["RTC_SKETCH_PACKED"]
        _reload_options()
    if k == 21:
        _check(ctx, oracle, seq, off, k, size=3400)


def test_packed_sketch_runs_that_touch_and_batch_edges(ctx, oracle):
    """runs that touch (what the packer emits at its seams), a run that ends the batch, a genome that starts the batch
    without 32 bases in front, genomes that end inside a lane's 64 bases, and an empty run list"""
    rng = np.random.default_rng(77)
    seq, off = _random_genomes(rng, [70_000, 4096 * 3 + 17, 40_000])
    seq[100:164] = ord("N")
    seq[-30:] = ord("N")
    packed, n_bases, runs = pack_batch(seq)
    r = runs.reshape(-1, 2)
    split = np.array([[100, 20], [120, 44]] + [list(x) for x in r[1:]], dtype=np.int64)  # the first run as two that touch
    want = oracle.sketch_minhash_batch(seq, off, 21, 1000)
    d_p = torch.from_numpy(packed).to(ctx.device)
    for rr in (r, split):
        sk = ctx.sketch_minhash_packed(d_p, off, k=21, size=1000, n_bases=n_bases, runs=torch.from_numpy(rr.reshape(-1).copy()).to(ctx.device))
        ctx.sync()
        for g, (a, b) in enumerate(zip(sk.to_host(), want)):
            assert np.array_equal(a, b), g
    clean, off2 = _random_genomes(rng, [64 * 1000])  # no run at all: the padding behind the genome is outside its extent
    packed2, nb2, runs2 = pack_batch(clean)
    sk = ctx.sketch_minhash_packed(torch.from_numpy(packed2).to(ctx.device), off2, k=21, size=1000, n_bases=nb2, runs=None)
    ctx.sync()
    assert np.array_equal(sk.to_host()[0], oracle.sketch_minhash_batch(clean, off2, 21, 1000)[0])


SOAK_SEEDS = int(os.environ.get("RTC_SOAK_SEEDS", "3"))


@pytest.mark.parametrize("seed", list(range(1, SOAK_SEEDS + 1)))
def test_packed_sketch_random_layouts(ctx, oracle, seed):
    """test_sketch_random_layouts' cases through the staging format: hundreds of genomes at arbitrary offsets, runs of every
    length placed at random, lower case, per-genome sketch sizes; k and the size regime per seed"""
    rng = np.random.default_rng(4000 + seed)
    k = [21, 17, 16, 32, 25, 19, 11, 28, 23, 31, 20, 13][(seed - 1) % 12]
    parts, off = [], [0]
    for g in range(200):
        L = int(rng.choice([0, 5, k - 1, k, k + 1, 63, 64, 65, 1000, 4095, 4096, 4097, 20_000, 60_000]))
        s = rng.choice(ACGT, size=L)
        for _ in range(int(rng.integers(0, 6)) if L else 0):
            a = int(rng.integers(0, L))
            ln = int(rng.choice([1, 1, 2, 3, 17, 64, 200, L]))
            s[a:a + ln] = rng.choice(np.frombuffer(b"NnRYKM-*\n", dtype=np.uint8), size=len(s[a:a + ln]))
        low = rng.random(L) < 0.2
        s[low & (s > 64)] |= 0x20
        parts.append(s)
        off.append(off[-1] + L)
    seq = np.concatenate(parts)
    off = np.array(off, dtype=np.uint64)
    if seed % 2:
        _check(ctx, oracle, seq, off, k, size=int(rng.choice([10, 100, 1000])))
    else:
        sizes = rng.integers(1, 1500, size=len(off) - 1).astype(np.uint32)
        _check(ctx, oracle, seq, off, k, sizes=sizes)
