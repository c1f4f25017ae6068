"""GPU parity of the inverted-join candidate-edge path (rtc_pairs_join.hip): forced on, its (i, j, common)
triples must equal the oracle's pairwise counts filtered by the reference's rules (src/MST.cpp:1468-1487)
and the tiled kernel's output for the same tile, pair for pair."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reload_options():
    """the library reads its RTC_* switches when a context is created: every live context reads them again"""
    from rabbittclust_amd import api
    api.reload_all_options()


def _make_sketches(rng, n, smin, smax, pool_bits=20, dtype=np.uint64):
    out = []
    for _ in range(n):
        s = int(rng.integers(smin, smax + 1))
        v = np.unique(rng.integers(0, 1 << pool_bits, size=s * 2 + 2, dtype=np.uint64))[:s]
        v = (v * np.uint64(0x9E3779B97F4A7C15)) if dtype == np.uint64 else v
        out.append(np.sort(v.astype(dtype)))
    return out


def _edges(ctx, dev, row0, row1, col0, col1, radio, mode, cap=1 << 22):
    old = os.environ.get("RTC_PAIR_JOIN")
    os.environ["RTC_PAIR_JOIN"] = str(mode)
    _reload_options()
    try:
        e, m = ctx.pair_edges(dev, row0, row1, col0, col1, radio, cap)
    finally:
        if old is None:
            os.environ.pop("RTC_PAIR_JOIN", None)
            _reload_options()
        else:
            os.environ["RTC_PAIR_JOIN"] = old
            _reload_options()
    assert m <= cap
    a = e[:m].cpu().numpy().view(np.uint32).astype(np.int64)
    order = np.lexsort((a[:, 1], a[:, 0]))
    return a[order]


def _expected(oracle, sk, row0, row1, col0, col1, radio):
    out = []
    for i in range(row0, row1):
        for j in range(col0, min(col1, i)):
            c = oracle.common(sk[i], sk[j])
            if c == 0:
                continue
            if radio >= 0:
                mn, mx = min(len(sk[i]), len(sk[j])), max(len(sk[i]), len(sk[j]))
                if mx > radio * mn:
                    continue
            out.append((i, j, c))
    return np.array(out, dtype=np.int64).reshape(-1, 3)


@pytest.mark.parametrize("width", [8, 4])
def test_join_equals_oracle_and_tiled(ctx, oracle, width):
    from rabbittclust_amd import api
    rng = np.random.default_rng(100 + width)
    dt = np.uint64 if width == 8 else np.uint32
    sk = _make_sketches(rng, 150, 0, 260, pool_bits=12, dtype=dt)
    sk[3] = np.zeros(0, dtype=dt)
    sk[5] = sk[4].copy()
    sk[149] = sk[4].copy()
    sk[70] = np.zeros(0, dtype=dt)
    top = np.array([np.iinfo(dt).max], dtype=dt)  # the tiled table's EMPTY marker as a hash value
    sk[10] = np.concatenate([sk[10], top]) if len(sk[10]) == 0 or sk[10][-1] != top[0] else sk[10]
    sk[90] = np.concatenate([sk[90], top]) if len(sk[90]) == 0 or sk[90][-1] != top[0] else sk[90]
    dev = api.SketchSet.from_host(sk, ctx.device, width=width)
    for (r0, r1, c0, c1, radio) in [(1, 150, 0, 149, -1), (1, 150, 0, 149, 2), (37, 131, 5, 120, -1), (100, 150, 0, 149, 3),
                                    (1, 40, 20, 39, -1), (149, 150, 0, 149, -1)]:
        want = _expected(oracle, sk, r0, r1, c0, c1, radio)
        got = _edges(ctx, dev, r0, r1, c0, c1, radio, mode=2)
        assert np.array_equal(got, want), (r0, r1, c0, c1, radio)
        os.environ["RTC_JOIN_SEMI"] = "2"  # semi-join forced: columns keep only the hashes a row has
        _reload_options()
        try:
            assert np.array_equal(_edges(ctx, dev, r0, r1, c0, c1, radio, mode=2), want), (r0, r1, c0, c1, radio, "semi")
        finally:
            os.environ.pop("RTC_JOIN_SEMI", None)
            _reload_options()
        tiled = _edges(ctx, dev, r0, r1, c0, c1, radio, mode=0)
        assert np.array_equal(tiled, want)


def test_join_appends_after_existing_edges_and_counts_past_the_capacity(ctx, oracle):
    """Contract shared with the tiled kernel: edges go in at *count, every survivor is counted even when the
    list is full (the caller grows it to the exact need and repeats)."""
    import torch
    from rabbittclust_amd import api
    rng = np.random.default_rng(7)
    sk = _make_sketches(rng, 80, 50, 90, pool_bits=10)
    dev = api.SketchSet.from_host(sk, ctx.device)
    want = _expected(oracle, sk, 1, 80, 0, 79, -1)
    assert len(want) > 100
    os.environ["RTC_PAIR_JOIN"] = "2"
    _reload_options()
    try:
        cap = 40 + len(want) // 2
        edges = torch.full((cap, 3), -1, dtype=torch.int32, device=ctx.device)
        count = torch.tensor([40], dtype=torch.int64, device=ctx.device)
        ctx.check(ctx.lib.rtc_pair_edges_dev(ctx.h, api._t_ptr(dev.hashes), dev.width, api._t_ptr(dev.start), api._t_ptr(dev.len),
                                             dev.n, 1, 80, 0, 79, -1, api._t_ptr(edges), cap, api._t_ptr(count)))
    finally:
        os.environ.pop("RTC_PAIR_JOIN", None)
        _reload_options()
    assert int(count.item()) == 40 + len(want)
    e = edges.cpu().numpy()
    assert (e[:40] == -1).all() and (e[40:] != -1).all()
    got = {tuple(int(v) for v in r) for r in e[40:]}
    assert got <= {tuple(int(v) for v in r) for r in want}
    assert len(got) == cap - 40


def test_join_long_posting_lists(ctx, oracle):
    """Forty copies of one genome among unrelated ones: posting lists of length 40 (the wave-wide emit)."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(9)
    sk = _make_sketches(rng, 120, 80, 100, pool_bits=30)
    for g in range(10, 120, 3):
        sk[g] = sk[7].copy()
    sk[50] = sk[7][::2].copy()
    dev = api.SketchSet.from_host(sk, ctx.device)
    want = _expected(oracle, sk, 1, 120, 0, 119, 4)
    got = _edges(ctx, dev, 1, 120, 0, 119, 4, mode=2)
    assert np.array_equal(got, want)
    assert np.array_equal(_edges(ctx, dev, 1, 120, 0, 119, 4, mode=0), want)


@pytest.mark.parametrize("sharers,n,width", [(120, 400, 8), (300, 500, 8), (1500, 1700, 8), (7000, 7200, 8), (1500, 40000, 8),
                                              (300, 500, 4), (1500, 1700, 4), (900, 40000, 4)])
def test_join_column_tail_light_and_heavy_columns(ctx, sharers, n, width):
    """The join's second half counts a column's partners in LDS: up to 640 distinct partners in a wave's hash table, more
    in the heavy kernel's per-row counters (16 320 row ids per walk: the 40 000-genome case takes three).  `sharers` genomes
    spread over the set hold one common hash (the first of them then has sharers - 1 distinct partners) beside family hashes
    with multiplicities; every size gives the tiled kernel's triples, also appended behind edges that are already there."""
    import torch
    from rabbittclust_amd import api
    rng = np.random.default_rng(sharers + n + width)
    dt = np.uint64 if width == 8 else np.uint32
    lo_h, hi_h = (1 << 40, 1 << 62) if width == 8 else (1 << 10, 1 << 31)
    common = np.uint64(0x123456789ABCDEF if width == 8 else 0x1234567)
    fam = [np.unique(rng.integers(lo_h, hi_h, size=24, dtype=np.uint64)) for _ in range(max(40, n // 100))]
    holds = np.zeros(n, dtype=bool)
    holds[rng.choice(n, size=sharers, replace=False)] = True
    sk = []
    for g in range(n):
        own = rng.integers(lo_h, hi_h, size=int(rng.integers(4, 12)), dtype=np.uint64)
        f = fam[int(rng.integers(0, len(fam)))]
        parts = [own, f[rng.random(len(f)) < 0.7]]
        if holds[g]:
            parts.append(np.array([common], dtype=np.uint64))
        sk.append(np.unique(np.concatenate(parts)).astype(dt))
    dev = api.SketchSet.from_host(sk, ctx.device, width=width)
    cap = 1 << 26
    for (r0, r1, c0, c1, radio) in [(1, n, 0, n - 1, -1), (n // 3, n, 0, n - 1, 2)]:
        want = _edges(ctx, dev, r0, r1, c0, c1, radio, mode=0, cap=cap)
        assert len(want) >= (sharers - 1) * (sharers - 2) // 8
        got = _edges(ctx, dev, r0, r1, c0, c1, radio, mode=2, cap=cap)
        assert np.array_equal(got, want), (sharers, n, r0, radio)
        if r0 > 1:  # the semi-join in front of the sort (descriptors then lie in the kept hashes' layout)
            os.environ["RTC_JOIN_SEMI"] = "2"
            _reload_options()
            try:
                assert np.array_equal(_edges(ctx, dev, r0, r1, c0, c1, radio, mode=2, cap=cap), want), (sharers, n, "semi")
            finally:
                os.environ.pop("RTC_JOIN_SEMI", None)
                _reload_options()
    want = _edges(ctx, dev, 1, n, 0, n - 1, -1, mode=0, cap=cap)
    edges = torch.full((cap, 3), -1, dtype=torch.int32, device=ctx.device)
    count = torch.tensor([40], dtype=torch.int64, device=ctx.device)
    os.environ["RTC_PAIR_JOIN"] = "2"
    _reload_options()
    try:
        ctx.check(ctx.lib.rtc_pair_edges_dev(ctx.h, api._t_ptr(dev.hashes), dev.width, api._t_ptr(dev.start), api._t_ptr(dev.len),
                                             dev.n, 1, n, 0, n - 1, -1, api._t_ptr(edges), cap, api._t_ptr(count)))
    finally:
        os.environ.pop("RTC_PAIR_JOIN", None)
        _reload_options()
    m = int(count.item())
    assert m == 40 + len(want)
    e = edges[:m].cpu().numpy()
    assert (e[:40] == -1).all()
    a = e[40:].view(np.uint32).astype(np.int64)
    assert np.array_equal(a[np.lexsort((a[:, 1], a[:, 0]))], want)


@pytest.mark.parametrize("kind", ["minhash", "kssd"])
def test_join_on_real_sketches_row_shards_equal_the_tiled_kernel(ctx, kind):
    """2 000 synthetic genomes in families: the default dispatch (cost rule), the forced join and the tiled
    kernel agree on every row range of an 8-way split; the MST built on top is the same either way."""
    from rabbittclust_amd import api, host, pipeline
    desc = api.synth_family_descs(200, 10, global_seed=5)
    L = 200_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    if kind == "minhash":
        sk = ctx.sketch_minhash(seq, off, k=21, size=1000)
    else:
        sk = ctx.sketch_kssd(seq, off, host.generate_shuffle_dim(6), kmer_size=21, drlevel=3)
    n = sk.n
    radio = 4
    total = 0
    bnd = pipeline.triangle_row_ranges(n, 8, fixed_cols=1.84 * 1000)
    for a, b in zip(bnd[:-1], bnd[1:]):
        a = max(a, 1)
        if a >= b:
            continue
        t = _edges(ctx, sk, a, b, 0, b - 1, radio, mode=0)
        j = _edges(ctx, sk, a, b, 0, b - 1, radio, mode=2)
        d = _edges(ctx, sk, a, b, 0, b - 1, radio, mode=1)
        assert np.array_equal(t, j) and np.array_equal(t, d)
        # the semi-join in front of the sort (columns keep only hashes some row has): forced on, and off
        for semi in ("2", "0"):
            os.environ["RTC_JOIN_SEMI"] = semi
            _reload_options()
            try:
                assert np.array_equal(_edges(ctx, sk, a, b, 0, b - 1, radio, mode=2), t), (a, b, semi)
            finally:
                os.environ.pop("RTC_JOIN_SEMI", None)
                _reload_options()
        total += len(t)
    assert total > n
    os.environ["RTC_PAIR_JOIN"] = "0"
    _reload_options()
    try:
        m0 = ctx.mst(sk, 0.05)
    finally:
        os.environ["RTC_PAIR_JOIN"] = "2"
        _reload_options()
    try:
        m1 = ctx.mst(sk, 0.05)
    finally:
        os.environ.pop("RTC_PAIR_JOIN", None)
        _reload_options()
    assert np.array_equal(m0, m1)


@pytest.mark.parametrize("prefixes", [400, 1])
def test_join_u64_hashes_that_share_their_upper_half(ctx, oracle, prefixes):
    """The join sorts 64-bit hashes on their 32 most significant varying bits and repairs the runs in which distinct
    hashes share them.  Hashes below 2^60 here: bits [28, 60) are the sorted ones.  400 prefixes: hundreds of mixed
    runs of a few dozen elements (repaired in place); one prefix: a single run of everything (the repair gives up,
    the sort is repeated on all bits)."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(21 + prefixes)
    pre = rng.integers(0, 1 << 32, size=prefixes, dtype=np.uint64)
    low = rng.integers(0, 1 << 28, size=64 if prefixes > 1 else 3000, dtype=np.uint64)
    sk = []
    for _ in range(130):
        s = int(rng.integers(60, 140))
        v = (pre[rng.integers(0, prefixes, size=s)] << np.uint64(28)) | low[rng.integers(0, len(low), size=s)]
        sk.append(np.unique(v))
    sk[11] = sk[10].copy()
    dev = api.SketchSet.from_host(sk, ctx.device)
    want = _expected(oracle, sk, 1, 130, 0, 129, -1)
    assert len(want) > 200
    got = _edges(ctx, dev, 1, 130, 0, 129, -1, mode=2)
    assert np.array_equal(got, want)
    os.environ["RTC_JOIN_FULLSORT"] = "1"
    _reload_options()
    try:
        assert np.array_equal(_edges(ctx, dev, 1, 130, 0, 129, -1, mode=2), want)
    finally:
        os.environ.pop("RTC_JOIN_FULLSORT", None)
        _reload_options()


def test_cost_rule_takes_the_join_for_families_of_forty_and_the_tiled_kernel_for_families_of_four_hundred(ctx):
    """The rule that picks the pair phase's path (RTC_PAIR_JOIN unset): 10 000 MinHash sketches of 1 000 hashes in families of
    40 near relatives go through the inverted join (its column kernel counts ~1e11 co-occurrences a second: 1.3 ms against
    the tiled kernel's 1.9), families of 400 through the tiled kernel (refused by the density sample, before any sort);
    either way the edges are the other path's."""
    from rabbittclust_amd import api
    L = 100_000
    for fam, path in ((40, 3), (400, 2)):
        desc = api.synth_family_descs(10000 // fam, fam, global_seed=7)
        off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
        sk = ctx.sketch_minhash(ctx.synth_genomes(desc, off), off, k=21, size=1000)
        n = sk.n
        got = _edges(ctx, sk, 1, n, 0, n - 1, 4, mode=1, cap=1 << 25)
        assert ctx.pair_last_path() == path, (fam, ctx.pair_last_path())
        other = _edges(ctx, sk, 1, n, 0, n - 1, 4, mode=0 if path == 3 else 2, cap=1 << 25)
        assert ctx.pair_last_path() == (2 if path == 3 else 3)
        assert np.array_equal(got, other), fam


@pytest.mark.parametrize("seed", list(range(1, int(os.environ.get("RTC_SOAK_SEEDS", "3")) + 1)))
def test_join_equals_tiled_on_mid_scale_random_families(ctx, seed):
    """A few thousand to thirty thousand small sketches in families of random size (1 .. 1 500 members: partner lists from one to
    over a thousand entries, columns on both sides of the 640-partner table limit, several columns per wave beyond 24 576
    columns), u64 or u32, random tiles and size filters, the semi-join forced now and then: the join gives the tiled kernel's
    triples."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(31000 + seed)
    width = 8 if seed % 2 else 4
    dt = np.uint64 if width == 8 else np.uint32
    lo_h, hi_h = (1 << 40, 1 << 62) if width == 8 else (1 << 8, 1 << 31)
    n = int(rng.choice([3000, 6000, 30000]))
    sk, g = [], 0
    while g < n:
        members = int(min(n - g, rng.choice([1, 2, 5, 20, 90, 700, 1500], p=[0.3, 0.2, 0.2, 0.15, 0.1, 0.04, 0.01])))
        core = np.unique(rng.integers(lo_h, hi_h, size=int(rng.integers(3, 30)), dtype=np.uint64))
        keep_p = float(rng.choice([0.3, 0.8, 1.0]))
        for _ in range(members):
            own = rng.integers(lo_h, hi_h, size=int(rng.integers(0, 6)), dtype=np.uint64)
            sk.append(np.unique(np.concatenate([own, core[rng.random(len(core)) < keep_p]])).astype(dt))
        g += members
    order = rng.permutation(n)  # families scattered over the ids
    sk = [sk[i] for i in order]
    dev = api.SketchSet.from_host(sk, ctx.device, width=width)
    cap = 1 << 25
    for t in range(3):
        r0 = 1 if t == 0 else int(rng.integers(1, n)); r1 = n if t == 0 else int(rng.integers(r0 + 1, n + 1))
        c0 = 0 if t == 0 else int(rng.integers(0, r1 - 1)); c1 = r1 - 1 if t == 0 else int(rng.integers(c0 + 1, r1))
        radio = int(rng.choice([-1, 2, 4]))
        want = _edges(ctx, dev, r0, r1, c0, c1, radio, mode=0, cap=cap)
        assert np.array_equal(_edges(ctx, dev, r0, r1, c0, c1, radio, mode=2, cap=cap), want), (seed, r0, r1, c0, c1, radio)
        if t == 2:
            os.environ["RTC_JOIN_SEMI"] = "2"
            _reload_options()
            try:
                assert np.array_equal(_edges(ctx, dev, r0, r1, c0, c1, radio, mode=2, cap=cap), want), (seed, r0, r1, c0, c1, radio, "semi")
            finally:
                os.environ.pop("RTC_JOIN_SEMI", None)
                _reload_options()


SOAK_SEEDS = int(os.environ.get("RTC_SOAK_SEEDS", "3"))  # RTC_SOAK_SEEDS=40: a longer walk through random tiles


@pytest.mark.parametrize("seed", list(range(1, SOAK_SEEDS + 1)))
def test_pair_paths_on_random_tiles(ctx, oracle, seed):
    """Random sketch sets (u64 / u32, ragged sizes 0 .. 600, hash pools from dense to sparse, duplicates of whole sketches,
    the table's empty marker as a hash), random row / column ranges and size-ratio filters: the join, the join with its
    semi-join forced and the tiled kernel all give the oracle's (i, j, common) triples."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(7000 + seed)
    dt = np.uint64 if seed % 2 else np.uint32
    n = int(rng.integers(40, 260))
    sk = _make_sketches(rng, n, 0, int(rng.choice([8, 60, 300, 600])), pool_bits=int(rng.choice([8, 11, 14, 20])), dtype=dt)
    for _ in range(int(rng.integers(0, 6))):
        a, b = rng.integers(0, n, size=2)
        sk[int(a)] = sk[int(b)].copy()
    if rng.random() < 0.5:
        q = int(rng.integers(0, n))
        top = np.array([np.iinfo(dt).max], dtype=dt)
        if len(sk[q]) == 0 or sk[q][-1] != top[0]:
            sk[q] = np.concatenate([sk[q], top])
    dev = api.SketchSet.from_host(sk, ctx.device, width=8 if dt == np.uint64 else 4)
    for _ in range(3):
        r0 = int(rng.integers(1, n)); r1 = int(rng.integers(r0 + 1, n + 1))
        c0 = int(rng.integers(0, r1 - 1)); c1 = int(rng.integers(c0 + 1, r1))
        radio = int(rng.choice([-1, 1, 2, 4]))
        want = _expected(oracle, sk, r0, r1, c0, c1, radio)
        assert np.array_equal(_edges(ctx, dev, r0, r1, c0, c1, radio, mode=2), want), (seed, r0, r1, c0, c1, radio, "join")
        os.environ["RTC_JOIN_SEMI"] = "2"
        _reload_options()
        try:
            assert np.array_equal(_edges(ctx, dev, r0, r1, c0, c1, radio, mode=2), want), (seed, r0, r1, c0, c1, radio, "semi")
        finally:
            os.environ.pop("RTC_JOIN_SEMI", None)
            _reload_options()
        assert np.array_equal(_edges(ctx, dev, r0, r1, c0, c1, radio, mode=0), want), (seed, r0, r1, c0, c1, radio, "tiled")


@pytest.mark.parametrize("width", [8, 4])
@pytest.mark.parametrize("shape", ["even", "crowded", "two_values", "large"])
def test_join_on_key_distributions_that_stress_its_sort(ctx, oracle, width, shape):
    """Key distributions that stress the sort in front of the join's count: hashes spread evenly, hashes CROWDED into a sliver
    below the largest one (the radix passes over the bits that vary see one digit), two distinct values (posting lists of
    thousands) and a set of 2.4 million records.  For u64 sets the sort runs on half the key bits and repairs the runs of distinct
    hashes that agree in them: `crowded` is nothing but such runs.  Forced join == tiled kernel, whole triangle, a row shard
    (semi-join in front of the sort) and a column window; the first 300 sketches against the oracle."""
    from rabbittclust_amd import api
    rng = np.random.default_rng({"even": 1, "crowded": 2, "two_values": 3, "large": 4}[shape] * 10 + width)
    dt = np.uint64 if width == 8 else np.uint32
    hi = 62 if width == 8 else 31
    if shape == "even":
        n, s = 3000, 60
        pool = np.unique(rng.integers(1, 1 << hi, size=40000, dtype=np.uint64)).astype(dt)
    elif shape == "crowded":
        n, s = 3000, 60
        pool = np.unique(np.concatenate([(np.uint64(1) << np.uint64(hi)) + rng.integers(0, 1 << 16, size=30000, dtype=np.uint64),
                                         rng.integers(1, 1 << 20, size=30, dtype=np.uint64)])).astype(dt)
    elif shape == "two_values":
        n, s = 2500, 2
        pool = np.array([7, (1 << hi) + 12345], dtype=dt)
    else:
        n, s = 12000, 200
        pool = np.unique(rng.integers(1, 1 << hi, size=600000, dtype=np.uint64)).astype(dt)
    sk = [np.sort(rng.choice(pool, size=min(s, len(pool)), replace=False)) for _ in range(n)]
    dev = api.SketchSet.from_host(sk, ctx.device, width=width, kind="kssd" if width == 4 else "minhash")
    cap = 1 << 23
    tiled = _edges(ctx, dev, 1, n, 0, n - 1, -1, 0, cap=cap)
    d0 = ctx.diag()
    own = _edges(ctx, dev, 1, n, 0, n - 1, -1, 2, cap=cap)
    assert ctx.diag()["join_tiles"] == d0["join_tiles"] + 1, "the join did not run"
    assert len(tiled) > 0 and np.array_equal(own, tiled)
    r0 = n - n // 8
    assert np.array_equal(_edges(ctx, dev, r0, n, 0, n - 1, -1, 2, cap=cap), _edges(ctx, dev, r0, n, 0, n - 1, -1, 0, cap=cap))
    assert np.array_equal(_edges(ctx, dev, n // 2, n, 100, n // 3, -1, 2, cap=cap), _edges(ctx, dev, n // 2, n, 100, n // 3, -1, 0, cap=cap))
    if shape != "large":
        want = _expected(oracle, sk[:300], 1, 300, 0, 299, -1)
        got = _edges(ctx, api.SketchSet.from_host(sk[:300], ctx.device, width=width), 1, 300, 0, 299, -1, 2)
        assert np.array_equal(got, want)
