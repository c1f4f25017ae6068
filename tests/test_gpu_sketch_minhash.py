"""GPU parity: MinHash sketch kernel vs the CPU oracle (bit-exact hash sets)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reload_options():
    """the library reads its RTC_* switches when a context is created: every live context reads them again"""
    from rabbittclust_amd import api
    api.reload_all_options()


def _random_genomes(rng, lens, n_rate=0.0, lower_rate=0.0):
    parts, off = [], [0]
    for L in lens:
        g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
        if n_rate and L:
            idx = rng.random(L) < n_rate
            g[idx] = rng.choice(np.frombuffer(b"NRYKMnx-", dtype=np.uint8), size=int(idx.sum()))
        if lower_rate and L:
            idx = rng.random(L) < lower_rate
            g[idx] |= 0x20
        parts.append(g)
        off.append(off[-1] + L)
    seq = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return seq, np.array(off, dtype=np.uint64)


def _check(ctx, oracle, seq, off, k, size=None, sizes=None):
    d = ctx.upload_sequences(seq)
    sk = ctx.sketch_minhash(d, off, k=k, size=size if size else 1, sizes=sizes)
    ctx.sync()
    got = sk.to_host()
    want = oracle.sketch_minhash_batch(seq, off, k, sizes if sizes is not None else size)
    assert len(got) == len(want)
    for g, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), f"genome {g}: k={k} got {len(a)} want {len(b)}"


@pytest.mark.parametrize("k", [21, 17, 16, 11, 32, 5, 18, 19, 20, 22, 23, 24, 28, 29, 1, 3, 9, 13, 25, 27, 31])
def test_sketch_matches_oracle_various_k(ctx, oracle, k):
    rng = np.random.default_rng(100 + k)
    seq, off = _random_genomes(rng, [50_000, 123_457, 80_001, 15_359, 15_361, 30_720])
    _check(ctx, oracle, seq, off, k, size=1000)


def test_sketch_with_n_runs_and_lowercase(ctx, oracle):
    rng = np.random.default_rng(7)
    seq, off = _random_genomes(rng, [200_000, 100_000, 60_000], n_rate=0.003, lower_rate=0.3)
    _check(ctx, oracle, seq, off, 21, size=1000)


def test_sketch_short_empty_and_ragged(ctx, oracle):
    rng = np.random.default_rng(8)
    lens = [0, 1, 20, 21, 22, 100, 0, 999, 5000, 16, 15375, 3]
    seq, off = _random_genomes(rng, lens)
    _check(ctx, oracle, seq, off, 21, size=1000)
    _check(ctx, oracle, seq, off, 21, size=50)


def test_sketch_multi_record_separator(ctx, oracle):
    rng = np.random.default_rng(9)
    recs = [rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L) for L in (5000, 30, 20, 7000, 21)]
    sep = np.frombuffer(b"\n", dtype=np.uint8)
    g = np.concatenate([x for r in recs for x in (r, sep)])
    seq = np.concatenate([g, g[::-1].copy()])
    off = np.array([0, len(g), 2 * len(g)], dtype=np.uint64)
    _check(ctx, oracle, seq, off, 21, size=400)
    # the oracle's own per-record update() must agree with the separator convention
    import ctypes as C
    oracle.lib().orc_mh_new.restype = C.c_void_p
    m = oracle.lib().orc_mh_new(21, 400, 42)
    for r in recs:
        oracle.lib().orc_mh_update(C.c_void_p(m), r.ctypes.data_as(C.c_void_p), C.c_uint64(len(r)))
    out = np.zeros(400, dtype=np.uint64)
    oracle.lib().orc_mh_store.restype = C.c_uint32
    c = oracle.lib().orc_mh_store(C.c_void_p(m), out.ctypes.data_as(C.c_void_p), C.c_uint32(400))
    want = oracle.sketch_minhash_batch(g, np.array([0, len(g)], dtype=np.uint64), 21, 400)[0]
    assert np.array_equal(out[:c], want)


def test_sketch_variable_sizes_containment_mode(ctx, oracle):
    rng = np.random.default_rng(10)
    lens = [400_000, 150_000, 90_000, 1_000_000]
    seq, off = _random_genomes(rng, lens)
    sizes = np.array([max(L // 200, 100) for L in lens], dtype=np.uint32)  # fileBytes/compress
    _check(ctx, oracle, seq, off, 21, sizes=sizes)


def test_sketch_repetitive_genome(ctx, oracle):
    # heavy duplication: the same 3 kb unit repeated -> every k-mer recurs ~70 times
    rng = np.random.default_rng(11)
    unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=3000)
    g = np.tile(unit, 70)
    poly = np.full(50_000, ord("A"), dtype=np.uint8)
    seq = np.concatenate([g, poly])
    off = np.array([0, len(g), len(g) + len(poly)], dtype=np.uint64)
    _check(ctx, oracle, seq, off, 21, size=1000)


def test_sketch_large_genome_is_segmented(ctx, oracle):
    # one 6 Mbp genome in a tiny batch is split into segments and merged
    d = oracle.synth_genome(1234, 99, 300, 6_000_000)
    off = np.array([0, len(d)], dtype=np.uint64)
    _check(ctx, oracle, d, off, 21, size=1000)


def test_synth_device_matches_oracle(ctx, oracle):
    from rabbittclust_amd import api
    desc = api.synth_family_descs(3, 3, global_seed=5, n_every=0)
    desc["n_every"][4] = 5000
    lens = [10_000, 33_333, 16, 70_001, 15, 60_000, 1, 100, 4097]
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    seq = ctx.synth_genomes(desc, off).cpu().numpy()
    for g, L in enumerate(lens):
        ref = oracle.synth_genome(int(desc[g]["fam_seed"]), int(desc[g]["mut_seed"]),
                                  int(desc[g]["mut_thr"]), L, int(desc[g]["n_every"]))
        assert np.array_equal(ref, seq[int(off[g]):int(off[g + 1])]), g


def test_sketch_size_beyond_one_lds_pass(ctx, oracle):
    """Sketch sizes above 6144 are selected in passes over ascending hash ranges: same bottom-s set."""
    rng = np.random.default_rng(31)
    # short (exhausted in pass 0), exhausted mid-pass, exactly-full, long single-segment and multi-segment genomes
    lens = [3000, 6164, 9000, 12308, 40_000, 700_000, 0, 25]
    seq, off = _random_genomes(rng, lens, n_rate=0.001, lower_rate=0.1)
    _check(ctx, oracle, seq, off, 21, size=8000)
    _check(ctx, oracle, seq, off, 21, size=6145)
    _check(ctx, oracle, seq, off, 21, size=12288)
    _check(ctx, oracle, seq, off, 17, size=20000)
    sizes = np.array([100, 7000, 20000, 6144, 6145, 13000, 9000, 8000], dtype=np.uint32)
    _check(ctx, oracle, seq, off, 21, sizes=sizes)


def test_sketch_large_size_multi_segment(ctx, oracle):
    """One large genome split into segments, several passes, partial-sketch merge per pass."""
    rng = np.random.default_rng(32)
    seq, off = _random_genomes(rng, [6_000_000, 50_000], n_rate=0.0005)
    _check(ctx, oracle, seq, off, 21, size=15000)
    # low-complexity genome: few distinct k-mers, exhausted before the last pass
    rep = np.tile(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=7001), 300)
    off2 = np.array([0, len(rep)], dtype=np.uint64)
    _check(ctx, oracle, rep, off2, 21, size=20000)


def test_sketch_stride_smaller_than_size_fails_loudly(ctx):
    from rabbittclust_amd import _lib
    import torch
    seq = np.frombuffer(b"ACGT" * 1000, dtype=np.uint8)
    d = ctx.upload_sequences(seq)
    off = np.array([0, len(seq)], dtype=np.uint64)
    out = torch.empty(100, dtype=torch.int64, device=ctx.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    st = ctx.lib.rtc_sketch_minhash_dev(ctx.h, d.data_ptr(), off.ctypes.data, 1, 21, 42, None, 1000,
                                        out.data_ptr(), 100, cnt.data_ptr())
    assert st == _lib.RTC_ERR_ARG


@pytest.mark.parametrize("size", [2035, 3318, 3319, 3574, 3575])
def test_sketch_sizes_around_the_third_workgroup_per_cu(ctx, oracle, size):
    """safe mode appends one k-mer per lane between two looks at the candidate count, so a third workgroup per CU fits up
    to s = 3 318 (3 574 with the packed tables): sizes on both sides of every boundary, on genomes that start without a
    threshold (short ones: everything passes at first, safe mode, merges under the minimum room)"""
    rng = np.random.default_rng(900 + size)
    seq, off = _random_genomes(rng, [300_000, 40_000, 3000, 1_200_000, 70_001], n_rate=0.0005)
    _check(ctx, oracle, seq, off, 21, size=size)


def test_sketch_many_tiny_genomes_and_max_size(ctx, oracle):
    rng = np.random.default_rng(12)
    lens = [int(x) for x in rng.integers(0, 3000, size=600)]
    lens[5] = 0
    seq, off = _random_genomes(rng, lens, n_rate=0.01, lower_rate=0.2)
    _check(ctx, oracle, seq, off, 21, size=1000)
    _check(ctx, oracle, seq, off, 32, size=64)
    # largest sketch size the LDS-resident selection takes (6144), on a genome with fewer k-mers than that
    seq2, off2 = _random_genomes(rng, [4000, 250_000])
    _check(ctx, oracle, seq2, off2, 21, size=6144)
    _check(ctx, oracle, seq2, off2, 17, size=6144)


def test_starting_threshold_restart_on_low_complexity(ctx, oracle):
    """Whole-genome workgroups start from a threshold 8x the expected s-th smallest hash and must run
    again from "none" when a genome has too few distinct k-mers below it: periodic genomes (period 3 000 and
    40 -- fewer than s distinct k-mers in all), a poly-A run, and normal genomes beside them; enough
    genomes in the batch that each is one workgroup (no partial segments)."""
    rng = np.random.default_rng(61)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts = []
    for g in range(3400):  # > 3 x 768 workgroup slots x ... -> whole-genome segments
        if g % 850 == 0:
            motif = rng.choice(acgt, size=3000)
            parts.append(np.tile(motif, 70)[:200_000])
        elif g % 850 == 1:
            parts.append(np.tile(rng.choice(acgt, size=40), 5000))
        elif g % 850 == 2:
            parts.append(np.full(150_000, ord("A"), dtype=np.uint8))
        else:
            parts.append(rng.choice(acgt, size=int(rng.integers(40_000, 60_000))))
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    seq = np.concatenate(parts)
    d = ctx.upload_sequences(seq)
    sk = ctx.sketch_minhash(d, off, k=21, size=1000)
    ctx.sync()
    got = sk.to_host()
    check = [g for g in range(len(parts)) if g % 850 < 4] + [5, 77, 1234, 3399]
    sub_off = np.zeros(len(check) + 1, dtype=np.uint64)
    sub_off[1:] = np.cumsum([len(parts[g]) for g in check])
    want = oracle.sketch_minhash_batch(np.concatenate([parts[g] for g in check]), sub_off, 21, 1000)
    for g, w in zip(check, want):
        assert np.array_equal(got[g], w), g
    assert len(got[0]) == 1000 and len(got[1]) < 100 and len(got[2]) == 1


def test_partial_segments_start_from_the_genome_threshold(ctx, oracle):
    """Few large genomes are cut into segments that all start from the genome's threshold; a genome with fewer than
    s distinct k-mers below it is flagged by the merge and walked again without one (second launch over the partial
    segments): a 3 Mbp periodic genome (period 3 000: fewer distinct k-mers than 3 s below T0), one of period 40 (fewer
    than s in all), a poly-A genome and normal ones beside them; also with the factor that flags about every second
    genome, with none, and with other segment counts -- identical sketches every way."""
    import os
    rng = np.random.default_rng(67)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts = [np.tile(rng.choice(acgt, size=3000), 1000),
             rng.choice(acgt, size=2_500_000),
             np.tile(rng.choice(acgt, size=40), 50_000),
             oracle.synth_genome(77, 5, 200, 3_200_000),
             np.full(1_700_000, ord("A"), dtype=np.uint8),
             rng.choice(acgt, size=1_900_000)]
    parts[3][1_000_000:1_000_050] = ord("N")
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(q) for q in parts])
    seq = np.concatenate(parts)
    want = oracle.sketch_minhash_batch(seq, off, 21, 1000)
    d = ctx.upload_sequences(seq)
    for env in ({}, {"RTC_SKETCH_T0_FACTOR": "1"}, {"RTC_SKETCH_T0_FACTOR": "0"}, {"RTC_SKETCH_ROUNDS": "1"},
                {"RTC_SKETCH_ROUNDS": "4", "RTC_SKETCH_T0_FACTOR": "2"}):
        os.environ.update(env)
        _reload_options()
        try:
            sk = ctx.sketch_minhash(d, off, k=21, size=1000)
            ctx.sync()
        finally:
            for key in env:
                del os.environ[key]
            _reload_options()
        got = sk.to_host()
        for g, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), (env, g, len(a), len(b))
    assert len(want[0]) == 1000 and len(want[2]) < 100 and len(want[4]) == 1


def test_starting_threshold_factor_does_not_change_results(ctx, oracle):
    """RTC_SKETCH_T0_FACTOR = 1 makes about half of the workgroups restart, 0 disables the starting
    threshold, 40 / 2000 let so many k-mers through that the express walk's per-wave queue fills up (the
    group is forgotten and the general walk takes over, then safe mode): identical sketches every way
    (and equal to the oracle on a sample)."""
    import os
    from rabbittclust_amd import api
    desc = api.synth_family_descs(120, 10, global_seed=9)
    L = 230_000  # six tiles: the inner four take the express walk
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    ref = ctx.sketch_minhash(seq, off, k=21, size=500)
    ctx.sync()
    for f in ("1", "0", "8", "40", "2000"):
        os.environ["RTC_SKETCH_T0_FACTOR"] = f
        _reload_options()
        try:
            alt = ctx.sketch_minhash(seq, off, k=21, size=500)
            ctx.sync()
        finally:
            del os.environ["RTC_SKETCH_T0_FACTOR"]
            _reload_options()
        import torch
        assert torch.equal(alt.hashes, ref.hashes) and torch.equal(alt.len, ref.len), f
    host = seq[: 3 * L].cpu().numpy()
    want = oracle.sketch_minhash_batch(host, off[:4], 21, 500)
    got = ref.to_host()
    assert all(np.array_equal(a, b) for a, b in zip(got[:3], want))


@pytest.mark.parametrize("k", [21, 17, 19, 23, 12, 28, 31, 32, 9])
def test_packed_table_layout_matches_oracle(ctx, oracle, k):
    """The packed LDS table layout (lo(b*c) in the entries' fourth dword): forced everywhere by RTC_SKETCH_PACKED, and
    picked by the launch on its own for sketch sizes it buys a third workgroup per CU for (k = 21: s = 3400)."""
    import os
    rng = np.random.default_rng(300 + k)
    seq, off = _random_genomes(rng, [260_000, 123_457, 15_361, 700_001], n_rate=0.0005, lower_rate=0.01)
    os.environ["RTC_SKETCH_PACKED"] = "1"
    _reload_options()
    try:
        _check(ctx, oracle, seq, off, k, size=1000)
    finally:
        del os.environ["RTC_SKETCH_PACKED"]
        _reload_options()
    if k == 21:
        _check(ctx, oracle, seq, off, k, size=3400)
        os.environ["RTC_SKETCH_NO_PACKED"] = "1"
        _reload_options()
        try:
            _check(ctx, oracle, seq, off, k, size=3400)
        finally:
            del os.environ["RTC_SKETCH_NO_PACKED"]
            _reload_options()


import os
SOAK_SEEDS = int(os.environ.get("RTC_SOAK_SEEDS", "3"))  # RTC_SOAK_SEEDS=60: a longer walk through random layouts


@pytest.mark.parametrize("seed", list(range(1, SOAK_SEEDS + 1)))
def test_sketch_random_layouts(ctx, oracle, seed):
    """Hundreds of genomes of 0 .. 60 000 bases at arbitrary offsets, runs of characters outside ACGT of every length placed at
    random (touching, ending a genome, covering one), lower case, per-genome sketch sizes; k and the size regime per seed."""
    rng = np.random.default_rng(4000 + seed)
    k = [21, 17, 16, 32, 25, 19, 11, 28, 23, 31, 20, 13][(seed - 1) % 12]
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts, off = [], [0]
    for g in range(200):
        L = int(rng.choice([0, 5, k - 1, k, k + 1, 63, 64, 65, 1000, 4863, 4864, 4865, 20_000, 60_000]))
        s = rng.choice(acgt, size=L)
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
        sizes = rng.integers(1, 1500, size=len(off) - 1).astype(np.uint32)   # containment mode: a size per genome
        _check(ctx, oracle, seq, off, k, sizes=sizes)
