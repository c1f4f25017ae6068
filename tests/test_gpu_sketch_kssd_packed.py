"""GPU parity: KSSD sketches straight from the 2-bit staging format (rtc_sketch_kssd_packed_dev) vs the oracle's
restatement of sketchFileWithKssd (src/SketchInfo.cpp:994-1193) run over the characters the batch was packed from.
Bit-exact sorted tuple lists; the packing below is the one the command lines' parser performs (base codes at 2 bits,
everything outside ACGT listed as runs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def pack_batch(seq):
    """characters -> (packed bytes padded to 64 bases, n_bases, runs[(start, length)]); lower case counts as its base"""
    n = len(seq)
    n_bases = (n + 63) // 64 * 64 + 64
    up = seq & 0xDF
    code = np.zeros(n_bases, dtype=np.uint8)
    valid = np.zeros(n_bases, dtype=bool)
    for c, ch in enumerate(b"ACGT"):
        m = up == ch
        code[:n][m] = c
        valid[:n] |= m
    b = code.reshape(-1, 4)
    packed = (b[:, 0] | (b[:, 1] << 2) | (b[:, 2] << 4) | (b[:, 3] << 6)).astype(np.uint8)
    bad = ~valid
    edge = np.diff(np.concatenate([[0], bad.astype(np.int8), [0]]))
    starts = np.nonzero(edge == 1)[0]
    ends = np.nonzero(edge == -1)[0]
    runs = np.stack([starts, ends - starts], axis=1).astype(np.int64).reshape(-1)
    return packed, n_bases, runs


def _sketch_packed(ctx, seq, off, sd, k, drlevel, stride=None):
    packed, n_bases, runs = pack_batch(seq)
    d_p = torch.from_numpy(packed).to(ctx.device)
    d_r = torch.from_numpy(runs).to(ctx.device)
    sk = ctx.sketch_kssd_packed(d_p, n_bases, d_r, off, sd, kmer_size=k, drlevel=drlevel, stride=stride)
    ctx.sync()
    return sk


def _genomes(rng, lens, n_rate=0.0):
    parts, off = [], [0]
    for L in lens:
        g = rng.choice(ACGT, size=L)
        if n_rate and L:
            idx = rng.random(L) < n_rate
            g[idx] = rng.choice(np.frombuffer(b"NnRy\n", dtype=np.uint8), size=int(idx.sum()))
        low = rng.random(L) < 0.1
        g[low] |= 0x20
        parts.append(g)
        off.append(off[-1] + L)
    return np.concatenate(parts), np.array(off, dtype=np.uint64)


@pytest.mark.parametrize("k,drlevel", [(21, 3), (22, 3), (19, 3), (21, 4), (25, 3), (17, 3), (23, 3), (27, 3), (18, 3), (28, 3)])
def test_packed_kssd_matches_oracle(ctx, oracle, k, drlevel):
    """Genomes that begin and end anywhere inside a lane's 64 bases (no run at the seams: the genome bounds alone keep
    k-mers from spanning them), scattered characters outside ACGT, lower case, an empty genome, one shorter than k."""
    rng = np.random.default_rng(k * 10 + drlevel)
    seq, off = _genomes(rng, [600_000, 250_001, 30_720, 10, 0, 123_456, 1_000_003], n_rate=0.001)
    p = oracle.kssd_params(k, drlevel)
    sd = oracle.kssd_shuffle_dim(p.half_subk)
    sk = _sketch_packed(ctx, seq, off, sd, k, drlevel)
    got = sk.to_host()
    assert sk.width == (8 if p.use64 else 4) and sk.k == p.kmer_size
    want = oracle.sketch_kssd_batch(seq, off, sd, kmer_size=k, drlevel=drlevel, threads=4)
    for g in range(len(off) - 1):
        assert np.array_equal(got[g], want[g]), f"genome {g} k={k} dr={drlevel}: got {len(got[g])} want {len(want[g])}"
    assert len(got[0]) > 50 or drlevel > 3


def test_packed_kssd_equals_the_ascii_kernel_on_a_large_batch(ctx, oracle):
    """Same tuples as rtc_sketch_kssd_dev over the unpacked batch: 40 genomes of 0.3-3 Mbp with N runs, many segments per
    genome and many chunks per wave."""
    rng = np.random.default_rng(5)
    lens = rng.integers(300_000, 3_000_000, size=40).tolist()
    seq, off = _genomes(rng, lens, n_rate=0.0002)
    for s in rng.integers(0, len(seq) - 5000, size=30):  # a few long runs
        seq[s:s + int(rng.integers(1, 5000))] = ord("N")
    sd = oracle.kssd_shuffle_dim(6)
    a = ctx.sketch_kssd(ctx.upload_sequences(seq), off, sd, kmer_size=21, drlevel=3).to_host()
    b = _sketch_packed(ctx, seq, off, sd, 21, 3).to_host()
    for g in range(len(lens)):
        assert np.array_equal(a[g], b[g]), g
    want = oracle.kssd_sketch(seq[int(off[3]):int(off[4])], 21, 3)
    assert np.array_equal(b[3], want)


@pytest.mark.parametrize("k", [21, 19, 27])
def test_packed_kssd_full_queues_contigs_and_edges(ctx, oracle, k):
    """A shuffle table that keeps the poly-A dimension (code 0 -- also what the packed stream holds under every run, so
    runs and gaps flood the candidate queues too), contigs cut by single characters every few hundred bases, genomes
    shorter than a lane's stretch, a wave's chunk, a k-mer."""
    rng = np.random.default_rng(100 + k)
    p = oracle.kssd_params(k, 3)
    sd = oracle.kssd_shuffle_dim(p.half_subk).copy()
    j = int(np.nonzero(sd == 7)[0][0])
    sd[0], sd[j] = sd[j], sd[0]

    def rnd(n):
        return rng.choice(ACGT, size=n)
    g0 = np.concatenate([rnd(150_000), np.full(30_000, ord("A"), np.uint8), rnd(50_000), np.full(9_000, ord("t"), np.uint8),
                         rnd(120_001), np.full(70_000, ord("N"), np.uint8), rnd(5_000)])
    g1 = rnd(400_000)
    cuts = np.sort(rng.choice(len(g1), size=900, replace=False))
    g1[cuts] = rng.choice(np.frombuffer(b"N>\n-", dtype=np.uint8), size=len(cuts))
    low = rng.random(len(g1)) < 0.3
    g1[low & (g1 > 64)] |= 0x20
    parts = [g0, g1, rnd(36_864 * 2), rnd(4_607), rnd(73), rnd(k), rnd(k - 2), np.zeros(0, np.uint8), rnd(250_000)]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in parts])
    seq = np.concatenate(parts)
    got = _sketch_packed(ctx, seq, off, sd, k, 3).to_host()
    want = oracle.sketch_kssd_batch(seq, off, sd, kmer_size=k, drlevel=3, threads=4)
    for g in range(len(parts)):
        assert np.array_equal(got[g], want[g]), f"genome {g} k={k}: got {len(got[g])} want {len(want[g])}"
    assert len(got[0]) > 50


@pytest.mark.parametrize("k", [21, 25])
def test_packed_kssd_stretches_that_flood_the_first_stage(ctx, oracle, k):
    """A genome stitched from kept 12-mers: a third of its dwords pass stage 1, the kernel leaves its steady-state loop
    in the middle of a lane's four words and walks the rest of the chunk dword by dword."""
    rng = np.random.default_rng(77 + k)
    p = oracle.kssd_params(k, 3)
    sd = np.asarray(oracle.kssd_shuffle_dim(p.half_subk))
    kept = np.nonzero((sd >= 0) & (sd < 4096))[0]

    def twelve(v):
        return ACGT[[(int(v) >> (2 * (11 - i))) & 3 for i in range(12)]]
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    pieces = []
    for v in rng.choice(kept, size=9000):
        m = twelve(v)
        if rng.random() < 0.5:
            m = np.array([comp[int(b)] for b in m[::-1]], dtype=np.uint8)
        pieces.append(m)
    flood = np.concatenate(pieces)
    rnd = lambda n: rng.choice(ACGT, size=n)
    g0 = np.concatenate([rnd(70_001), flood[:60_000], np.full(37, ord("N"), dtype=np.uint8), flood[60_000:], rnd(50_003)])
    g1 = rnd(150_000)
    seq = np.concatenate([g0, g1])
    off = np.array([0, len(g0), len(g0) + len(g1)], dtype=np.uint64)
    got = _sketch_packed(ctx, seq, off, sd, k, 3).to_host()
    for g in range(2):
        want = oracle.kssd_sketch(seq[int(off[g]):int(off[g + 1])], k, 3)
        assert np.array_equal(got[g], want), f"genome {g}: got {len(got[g])} want {len(want)}"
    assert len(got[0]) > 100


def test_packed_kssd_overflow_protocol_and_unsupported_configurations(ctx, oracle):
    rng = np.random.default_rng(6)
    g = np.tile(rng.choice(ACGT, size=40_000), 12)
    off = np.array([0, len(g)], dtype=np.uint64)
    sd = oracle.kssd_shuffle_dim(6)
    sk = _sketch_packed(ctx, g, off, sd, 21, 3, stride=16)   # RTC_ERR_OVERFLOW reports the stride, the wrapper retries
    assert np.array_equal(sk.to_host()[0], oracle.kssd_sketch(g, 21, 3))
    from rabbittclust_amd.api import RtcError
    for k, dr in [(31, 3), (15, 3), (21, 2)]:                # outside the prefilter kernel: callers unpack instead
        p = oracle.kssd_params(k, dr)
        with pytest.raises(RtcError) as e:
            _sketch_packed(ctx, g, off, oracle.kssd_shuffle_dim(p.half_subk), k, dr)
        assert "UNSUPPORTED" in str(e.value).upper()


import os
SOAK_SEEDS = int(os.environ.get("RTC_SOAK_SEEDS", "3"))  # RTC_SOAK_SEEDS=60: a longer walk through random layouts


@pytest.mark.parametrize("seed", list(range(1, SOAK_SEEDS + 1)))
def test_packed_kssd_random_layouts(ctx, oracle, seed):
    """Hundreds of genomes of 0 .. 40 000 bases at arbitrary offsets (no alignment to bytes, words or lanes), runs of every
    length from one character to whole genomes placed at random -- touching each other, ending a genome, covering one -- and
    gaps between genomes that belong to no genome at all; k and drlevel drawn per seed."""
    rng = np.random.default_rng(1000 + seed)
    k = [21, 18, 26, 17, 22, 23, 24, 25, 27, 28, 19, 20][(seed - 1) % 12]
    dr = [3, 3, 4][(seed - 1) % 3]
    parts, off_b, off_e, pos = [], [], [], 0
    for g in range(300):
        gap = int(rng.integers(0, 70)) if rng.random() < 0.5 else 0
        if gap:
            parts.append(rng.choice(ACGT, size=gap)); pos += gap   # bases between two genomes: owned by nobody
        L = int(rng.choice([0, 5, k - 1, k, k + 1, 63, 64, 65, 1000, 4095, 4096, 4097, 20_000, 40_000]))
        s = rng.choice(ACGT, size=L)
        nr = int(rng.integers(0, 6)) if L else 0
        for _ in range(nr):
            a = int(rng.integers(0, L))
            ln = int(rng.choice([1, 1, 2, 3, 17, 64, 200, L]))
            s[a:a + ln] = rng.choice(np.frombuffer(b"NnRYKM-*", dtype=np.uint8), size=len(s[a:a + ln]))
        low = rng.random(L) < 0.2
        s[low & (s > 64)] |= 0x20
        parts.append(s); off_b.append(pos); pos += L; off_e.append(pos)
    seq = np.concatenate(parts)
    # the API takes contiguous genomes (off[g + 1] is where g ends AND g + 1 begins): the gaps become genomes of their own
    bounds = sorted(set([0] + off_b + off_e + [len(seq)]))
    off = np.array(bounds, dtype=np.uint64)
    p = oracle.kssd_params(k, dr)
    sd = oracle.kssd_shuffle_dim(p.half_subk)
    got = _sketch_packed(ctx, seq, off, sd, k, dr).to_host()
    want = oracle.sketch_kssd_batch(seq, off, sd, kmer_size=k, drlevel=dr, threads=4)
    assert sum(len(w) for w in want) > (20 if dr == 3 else 2)
    for g in range(len(off) - 1):
        assert np.array_equal(got[g], want[g]), f"genome {g} [{off[g]}, {off[g + 1]}): got {len(got[g])} want {len(want[g])}"


def test_packed_kssd_without_any_run(ctx, oracle):
    """n_runs = 0 with a null run pointer: every base of the batch is ACGT, the padding behind the last genome is not owned."""
    rng = np.random.default_rng(9)
    seq = rng.choice(ACGT, size=64 * 5000)          # a multiple of 64: no padding characters at all inside the genomes
    off = np.array([0, 100_000, 100_000, len(seq)], dtype=np.uint64)
    packed, n_bases, runs = pack_batch(seq)
    assert len(runs) == 2 and runs[0] == len(seq)   # only the pad run; leave it out: nothing owns those bases
    d_p = torch.from_numpy(packed).to(ctx.device)
    sd = oracle.kssd_shuffle_dim(6)
    sk = ctx.sketch_kssd_packed(d_p, n_bases, None, off, sd, kmer_size=21, drlevel=3)
    got = sk.to_host()
    for g in range(3):
        assert np.array_equal(got[g], oracle.kssd_sketch(seq[int(off[g]):int(off[g + 1])], 21, 3))
