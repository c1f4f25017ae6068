"""GPU parity: KSSD (--fast) sketch kernel vs the oracle's restatement of sketchFileWithKssd
(src/SketchInfo.cpp:994-1193).  Bit-exact sorted tuple lists."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _genomes(rng, lens, n_rate=0.0):
    parts, off = [], [0]
    for L in lens:
        g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
        if n_rate and L:
            idx = rng.random(L) < n_rate
            g[idx] = rng.choice(np.frombuffer(b"NnRy\n", dtype=np.uint8), size=int(idx.sum()))
        low = rng.random(L) < 0.1
        g[low] |= 0x20
        parts.append(g)
        off.append(off[-1] + L)
    return np.concatenate(parts), np.array(off, dtype=np.uint64)


def _check(ctx, oracle, seq, off, k, drlevel):
    p = oracle.kssd_params(k, drlevel)
    sd = oracle.kssd_shuffle_dim(p.half_subk)
    d = ctx.upload_sequences(seq)
    sk = ctx.sketch_kssd(d, off, sd, kmer_size=k, drlevel=drlevel)
    ctx.sync()
    got = sk.to_host()
    assert sk.width == (8 if p.use64 else 4)
    assert sk.k == p.kmer_size
    for g in range(len(off) - 1):
        want = oracle.kssd_sketch(seq[int(off[g]):int(off[g + 1])], k, drlevel)
        assert np.array_equal(got[g], want), f"genome {g} k={k} dr={drlevel}: got {len(got[g])} want {len(want)}"
    return got


@pytest.mark.parametrize("k,drlevel", [(21, 3), (22, 3), (19, 3), (21, 4), (31, 3), (25, 3), (17, 3), (23, 3), (27, 3)])
def test_kssd_matches_oracle(ctx, oracle, k, drlevel):
    rng = np.random.default_rng(k * 10 + drlevel)
    seq, off = _genomes(rng, [600_000, 250_001, 30_720, 10, 0, 123_456], n_rate=0.001)
    got = _check(ctx, oracle, seq, off, k, drlevel)
    assert len(got[0]) > 50 or drlevel > 3


def test_kssd_hbm_table_path_drlevel2(ctx, oracle):
    # drlevel 2 keeps 65536 dimensions: the HBM table path instead of the LDS index
    rng = np.random.default_rng(5)
    seq, off = _genomes(rng, [200_000, 90_000])
    _check(ctx, oracle, seq, off, 21, 2)


def test_kssd_overflow_protocol_and_duplicates(ctx, oracle):
    # highly repetitive genome: many occurrences per kept tuple; tiny stride forces the retry path
    rng = np.random.default_rng(6)
    unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=40_000)
    g = np.tile(unit, 12)
    off = np.array([0, len(g)], dtype=np.uint64)
    p = oracle.kssd_params(21, 3)
    sd = oracle.kssd_shuffle_dim(p.half_subk)
    d = ctx.upload_sequences(g)
    sk = ctx.sketch_kssd(d, off, sd, kmer_size=21, drlevel=3, stride=16)
    got = sk.to_host()[0]
    want = oracle.kssd_sketch(g, 21, 3)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("k,drlevel,lens", [
    (21, 2, [12_000_000, 300_000, 20_000_000]),   # u32 tuples: 2 chunks (one merge pass, dedup from scratch) and 3 chunks
    (25, 2, [12_000_000, 50_000]),                # u64 tuples: 3 chunks of 16384 (two passes, in-place dedup)
    (21, 3, [140_000_000]),                       # LDS cuckoo index path with a row just above one LDS buffer
])
def test_kssd_rows_beyond_one_lds_buffer(ctx, oracle, k, drlevel, lens):
    """Genomes yielding more tuples than the in-LDS sort holds (32 768 u32 / 16 384 u64) go through
    the chunk-sort + merge-pass + dedup path; small rows of the same batch keep the LDS path."""
    from rabbittclust_amd import api
    desc = api.synth_family_descs(len(lens), 1, global_seed=900 + k + drlevel, n_every=0)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    d = ctx.synth_genomes(desc, off)
    p = oracle.kssd_params(k, drlevel)
    sd = oracle.kssd_shuffle_dim(p.half_subk)
    sk = ctx.sketch_kssd(d, off, sd, kmer_size=k, drlevel=drlevel)
    got = sk.to_host()
    seq = d.cpu().numpy()
    cap = 16384 if p.use64 else 32768
    assert max(len(x) for x in got) > cap
    for g in range(len(lens)):
        want = oracle.kssd_sketch(seq[int(off[g]):int(off[g + 1])], k, drlevel)
        assert np.array_equal(got[g], want), f"genome {g}: got {len(got[g])} want {len(want)}"


def test_kssd_big_row_with_heavy_duplication(ctx, oracle):
    """A repetitive genome: ~10x more appended tuples than distinct ones, all beyond one LDS buffer."""
    rng = np.random.default_rng(77)
    unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=3_000_000)
    g = np.tile(unit, 10)
    off = np.array([0, len(g)], dtype=np.uint64)
    p = oracle.kssd_params(21, 2)
    sd = oracle.kssd_shuffle_dim(p.half_subk)
    sk = ctx.sketch_kssd(ctx.upload_sequences(g), off, sd, kmer_size=21, drlevel=2)
    want = oracle.kssd_sketch(g, 21, 2)
    assert np.array_equal(sk.to_host()[0], want)


@pytest.mark.parametrize("k", [21, 19, 17, 27])
def test_kssd_full_queue_contigs_and_edges(ctx, oracle, k):
    """The steady-state walk's exits: (1) a shuffle table that keeps the dimension of the poly-A k-mer, so a
    poly-A stretch makes every lane a candidate at every position (the per-wave queue overflows, the group is
    forgotten and walked base by base with drains in between); (2) contigs separated by single characters
    outside ACGTacgt every few hundred bases; (3) genomes shorter than a tile, a wave's run, a k-mer."""
    rng = np.random.default_rng(100 + k)
    p = oracle.kssd_params(k, 3)
    sd = oracle.kssd_shuffle_dim(p.half_subk).copy()
    j = int(np.nonzero(sd == 7)[0][0])
    sd[0], sd[j] = sd[j], sd[0]  # dim_id 0 (AAAA...A / TTTT...T) is kept with rank 7
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n):
        return rng.choice(acgt, size=n)
    g0 = np.concatenate([rnd(150_000), np.full(30_000, ord("A"), np.uint8), rnd(50_000), np.full(9_000, ord("t"), np.uint8),
                         rnd(120_001)])
    g1 = rnd(400_000)
    cuts = np.sort(rng.choice(len(g1), size=900, replace=False))
    g1[cuts] = rng.choice(np.frombuffer(b"N>\n-", dtype=np.uint8), size=len(cuts))
    low = rng.random(len(g1)) < 0.3
    g1[low & (g1 > 64)] |= 0x20
    parts = [g0, g1, rnd(36_864 * 2), rnd(4_607), rnd(73), rnd(k), rnd(k - 2), np.zeros(0, np.uint8), rnd(250_000)]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in parts])
    seq = np.concatenate(parts)
    sk = ctx.sketch_kssd(ctx.upload_sequences(seq), off, sd, kmer_size=k, drlevel=3)
    ctx.sync()
    got = sk.to_host()
    want = oracle.sketch_kssd_batch(seq, off, sd, kmer_size=k, drlevel=3, threads=4)
    for g in range(len(parts)):
        assert np.array_equal(got[g], want[g]), f"genome {g} k={k}: got {len(got[g])} want {len(want[g])}"
    assert len(got[0]) > 50


@pytest.mark.parametrize("k", [21, 25])
def test_kssd_stretches_that_flood_the_first_stage(ctx, oracle, k):
    """A genome stitched from KEPT 12-mers (members of the shuffled dimension set and their reverse complements): about a
    third of its dwords pass the prefilter's first stage, more than the stage-1 queue takes per chunk, so the kernel
    walks those chunks dword by dword with the queues served in between; random stretches, an N run and a second
    ordinary genome sit around them.  Tuple lists must still equal the oracle's."""
    rng = np.random.default_rng(77 + k)
    p = oracle.kssd_params(k, 3)
    sd = np.asarray(oracle.kssd_shuffle_dim(p.half_subk))
    kept = np.nonzero((sd >= 0) & (sd < 4096))[0]
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)

    def twelve(v):  # 24-bit dim_id -> its 12 bases, first base in the top bits
        return alphabet[[(int(v) >> (2 * (11 - i))) & 3 for i in range(12)]]
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    pieces = []
    for v in rng.choice(kept, size=9000):
        m = twelve(v)
        if rng.random() < 0.5:
            m = np.array([comp[int(b)] for b in m[::-1]], dtype=np.uint8)
        pieces.append(m)
    flood = np.concatenate(pieces)                                  # 108 000 bases of back-to-back members
    rnd = lambda n: rng.choice(alphabet, size=n)
    g0 = np.concatenate([rnd(70_001), flood[:60_000], np.full(37, ord("N"), dtype=np.uint8), flood[60_000:], rnd(50_003)])
    g1 = rnd(150_000)
    seq = np.concatenate([g0, g1])
    off = np.array([0, len(g0), len(g0) + len(g1)], dtype=np.uint64)
    got = _check(ctx, oracle, seq, off, k, 3)
    assert len(got[0]) > 100
