"""GPU parity: all-pairs intersection counts vs the CPU oracle (exact integers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make_sketches(rng, n, smin, smax, pool_bits=20, dtype=np.uint64):
    """Random sorted distinct sets drawn from a small pool so intersections are frequent."""
    out = []
    for _ in range(n):
        s = int(rng.integers(smin, smax + 1))
        v = np.unique(rng.integers(0, 1 << pool_bits, size=s * 2, dtype=np.uint64))[:s]
        # spread into the full range but keep collisions between genomes
        v = (v * np.uint64(0x9E3779B97F4A7C15)) if dtype == np.uint64 else v
        out.append(np.sort(v.astype(dtype)))
    return out


def _oracle_matrix(oracle, sk):
    n = len(sk)
    m = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        for j in range(n):
            m[i, j] = oracle.common(sk[i], sk[j])
    return m


@pytest.mark.parametrize("algo", [1, 0])
@pytest.mark.parametrize("width", [8, 4])
def test_pair_common_matches_oracle(ctx, oracle, algo, width):
    from rabbittclust_amd import api
    rng = np.random.default_rng(width * 10 + algo)
    dt = np.uint64 if width == 8 else np.uint32
    sk = _make_sketches(rng, 70, 0, 300, pool_bits=12, dtype=dt)
    sk[3] = np.zeros(0, dtype=dt)
    sk[5] = sk[4].copy()
    dev = api.SketchSet.from_host(sk, ctx.device, width=width)
    got = ctx.pair_common(dev, algo=algo).cpu().numpy()
    want = _oracle_matrix(oracle, sk)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("algo", [1, 0])
def test_pair_common_subtile_and_lower(ctx, oracle, algo):
    from rabbittclust_amd import api
    rng = np.random.default_rng(3)
    sk = _make_sketches(rng, 150, 50, 120, pool_bits=11)
    dev = api.SketchSet.from_host(sk, ctx.device)
    want = _oracle_matrix(oracle, sk)
    got = ctx.pair_common(dev, row0=37, row1=131, col0=5, col1=150, algo=algo).cpu().numpy()
    assert np.array_equal(got, want[37:131, 5:150])
    low = ctx.pair_common(dev, lower_only=True, algo=algo).cpu().numpy()
    il = np.tril_indices(150, -1)
    assert np.array_equal(low[il], want[il])


def test_pair_common_on_real_sketches(ctx, oracle):
    from rabbittclust_amd import api
    desc = api.synth_family_descs(6, 5, global_seed=11)
    L = 120_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    dev = ctx.sketch_minhash(seq, off, k=21, size=1000)
    host = dev.to_host()
    want = _oracle_matrix(oracle, host)
    for algo in (1, 0):
        got = ctx.pair_common(dev, algo=algo).cpu().numpy()
        assert np.array_equal(got, want), algo
    assert want[1, 0] > 100 and want[5, 0] == 0  # families share hashes, strangers do not


def test_tiled_sentinel_values_and_variable_sizes(ctx, oracle):
    """Hash values equal to the table's EMPTY sentinel (all ones) and ragged sizes up to 5000."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(17)
    pool = np.unique(rng.integers(0, 1 << 63, size=30000, dtype=np.uint64))
    sk = []
    for g in range(130):
        size = int(rng.integers(1, 5000)) if g % 7 else int(rng.integers(0, 3))
        v = np.sort(rng.choice(pool, size=size, replace=False))
        if g % 3 == 0:
            v = np.append(v, np.uint64(0xFFFFFFFFFFFFFFFF))
        if g % 5 == 0 and len(v):
            v = np.insert(v, 0, np.uint64(0)) if v[0] != 0 else v
        sk.append(v.astype(np.uint64))
    dev = api.SketchSet.from_host(sk, ctx.device)
    want = _oracle_matrix(oracle, sk)
    got = ctx.pair_common(dev, algo=2).cpu().numpy()
    assert np.array_equal(got, want)


def test_tiled_u32_sentinel(ctx, oracle):
    from rabbittclust_amd import api
    rng = np.random.default_rng(18)
    sk = []
    for g in range(90):
        v = np.unique(rng.integers(0, 1 << 14, size=int(rng.integers(0, 700)), dtype=np.uint64)).astype(np.uint32)
        if g % 2:
            v = np.append(v, np.uint32(0xFFFFFFFF))
        sk.append(v)
    dev = api.SketchSet.from_host(sk, ctx.device, width=4)
    want = _oracle_matrix(oracle, sk)
    got = ctx.pair_common(dev, algo=2).cpu().numpy()
    assert np.array_equal(got, want)


def _colliding_pool(rng, width, want=2500):
    """Keys whose table coordinates collide in rtc_pairs_tiled.hip: the digest (image of a 32-bit key under x 0x9E3779B1, high
    half of the image of a 64-bit key under x 0x9E3779B97F4A7C15) picks the bucket with its top 12 bits and the fingerprint
    with bits 4..17.  Returns groups of >= 6 keys that agree in BOTH (more than a bucket's four slots: the bucket overflows
    and every lookup in the group is decided by the key comparison), plus keys that share only the bucket."""
    n = 1 << 26
    if width == 4:
        keys = np.unique(rng.integers(0, 1 << 32, size=n, dtype=np.uint64)).astype(np.uint64)
        dig = (keys * np.uint64(0x9E3779B1)) & np.uint64(0xffffffff)
    else:
        keys = np.unique(rng.integers(0, 1 << 63, size=n, dtype=np.uint64))
        dig = (keys * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)
    # one fingerprint (n / 2^14 keys), then the buckets that received the most of them
    sel = np.nonzero(((dig >> np.uint64(4)) & np.uint64(0x3fff)) == np.uint64(0x155))[0]   # one fingerprint: n / 2^14 keys
    sub, ssig = keys[sel], (dig[sel] >> np.uint64(20))
    order = np.argsort(ssig, kind="stable")
    sub, ssig = sub[order], ssig[order]
    uniq, start, cnt = np.unique(ssig, return_index=True, return_counts=True)
    pool = []
    for st, c in zip(start[np.argsort(-cnt)], np.sort(cnt)[::-1]):
        if c < 2 or len(pool) >= want:
            break
        pool.extend(sub[st:st + c].tolist())
    return np.array(sorted(set(pool)), dtype=np.uint64), int(np.max(cnt))


@pytest.mark.parametrize("width", [8, 4])
def test_tiled_fingerprint_collisions(ctx, oracle, width):
    """Hashes chosen so that they share fingerprint AND bucket in the tiled kernel's table (groups larger than a bucket):
    every probe of such a key finds its fingerprint in an overflowed bucket, all decisions fall to the comparison of the
    whole key image, and the walk has to follow the overflow into the next buckets; the sketches also overlap heavily
    (carry-save + ripple of the bit-sliced counters)."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(23 + width)
    pool, biggest = _colliding_pool(rng, width)
    assert len(pool) >= 1500 and biggest >= (3 if width == 8 else 3)
    filler = np.unique(rng.integers(0, 1 << (62 if width == 8 else 32), size=4000, dtype=np.uint64))
    pool = np.unique(np.concatenate([pool, filler]))
    sk = []
    for g in range(150):
        size = int(rng.integers(200, 3000))
        v = np.sort(rng.choice(pool, size=min(size, len(pool)), replace=False))
        sk.append(v.astype(np.uint64 if width == 8 else np.uint32))
    dev = api.SketchSet.from_host(sk, ctx.device, width=width)
    want = _oracle_matrix(oracle, sk)
    got = ctx.pair_common(dev, algo=2).cpu().numpy()
    assert np.array_equal(got, want)
    assert want[3, 2] > 10


def test_tiled_skewed_slices_force_row_subblocks(ctx, oracle):
    """All sketches crowd into a narrow value range except a few outliers that drag the sampled
    quantile boundaries: some (row block, partition) then holds far more keys than one table."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(19)
    sk = []
    n = 200
    for g in range(n):
        if g % 64 == 0:  # the sampled genomes (every n/64-th) are wide-range
            v = np.unique(rng.integers(0, 1 << 62, size=1500, dtype=np.uint64))
        else:            # everyone else lives in [2^40, 2^40 + 2^16)
            v = np.unique((1 << 40) + rng.integers(0, 1 << 16, size=900, dtype=np.uint64)).astype(np.uint64)
        sk.append(np.sort(v))
    dev = api.SketchSet.from_host(sk, ctx.device)
    want = _oracle_matrix(oracle, sk)
    got = ctx.pair_common(dev, algo=0).cpu().numpy()
    assert np.array_equal(got, want)
    low = ctx.pair_common(dev, row0=64, row1=190, col0=0, col1=189, lower_only=True, algo=0).cpu().numpy()
    for i in range(64, 190):
        assert np.array_equal(low[i - 64, :min(i, 189)], want[i, :min(i, 189)])


def test_tiled_matches_merge_kernel_at_scale(ctx):
    """Size-independent cross-check at a larger size: the two device algorithms must agree."""
    from rabbittclust_amd import api
    desc = api.synth_family_descs(150, 8, global_seed=23)
    L = 30_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    dev = ctx.sketch_minhash(seq, off, k=21, size=1000)
    a = ctx.pair_common(dev, lower_only=True, algo=1)
    b = ctx.pair_common(dev, lower_only=True, algo=2)
    il = np.tril_indices(dev.n, -1)
    assert np.array_equal(a.cpu().numpy()[il], b.cpu().numpy()[il])
    full = ctx.pair_common(dev, algo=2).cpu().numpy()
    assert np.array_equal(full, full.T)            # symmetry
    assert np.array_equal(np.diag(full), dev.len.cpu().numpy())  # |A ∩ A| = |A|


def test_tiled_large_sketches_match_merge_kernel(ctx):
    """Sketch sizes well beyond the default (multi-pass sketching, 20 000 hashes per genome): the tiled
    kernel must agree with the one-lane-per-pair merge kernel, and counts must be symmetric."""
    from rabbittclust_amd import api
    desc = api.synth_family_descs(13, 10, global_seed=29)
    L = 400_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    dev = ctx.sketch_minhash(seq, off, k=21, size=20000)
    assert int(dev.len.min()) == 20000
    a = ctx.pair_common(dev, algo=1).cpu().numpy()
    b = ctx.pair_common(dev, algo=2).cpu().numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(b, b.T) and np.array_equal(np.diag(b), dev.len.cpu().numpy())
    assert (b[np.triu_indices(dev.n, 1)] > 2000).sum() >= 13 * 45 // 2  # family members share a large part


def test_auto_dispatch_with_a_huge_sketch(ctx, oracle):
    """One sketch of 1.2 M hashes (a KSSD sketch of a multi-Gbp genome) among ordinary ones: algo 0
    must still return exact counts (falls back to the merge kernel when the tiled plan cannot take it)."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(41)
    sk = _make_sketches(rng, 40, 500, 1500, pool_bits=22, dtype=np.uint32)
    huge = np.unique(rng.integers(0, 1 << 22, size=1_500_000, dtype=np.uint64)).astype(np.uint32)[:1_200_000]
    sk[7] = np.sort(huge)
    dev = api.SketchSet.from_host(sk, ctx.device, width=4)
    got = ctx.pair_common(dev, algo=0).cpu().numpy()
    for i in (0, 7, 8, 39):
        for j in range(40):
            assert got[i, j] == oracle.common(sk[i], sk[j]) == got[j, i], (i, j)


def test_mash_union_truncated_estimator(ctx, oracle):
    """D3 (modifyMST's MinHash::distance()): Mash's estimator stops after `s` union elements, so it differs
    from the set-Jaccard counts of the index path; checked against the oracle restatement and two
    hand-computed cases (RabbitSketch itself is absent: parity unpinned)."""
    import ctypes as C
    from rabbittclust_amd import api
    rng = np.random.default_rng(23)
    pool = np.unique(rng.integers(1, 1 << 50, size=1200, dtype=np.uint64))
    sk = [np.sort(rng.choice(pool, size=int(rng.integers(30, 300)), replace=False)) for _ in range(70)]
    sk[3] = np.zeros(0, dtype=np.uint64)
    sk[5] = sk[4].copy()
    sk.append(np.array([1, 2, 3, 4, 5, 6], dtype=np.uint64))
    sk.append(np.array([2, 4, 6, 8, 10, 12], dtype=np.uint64))
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    n = len(sk)
    L = oracle.lib()
    for s in (6, 100, 1000):
        common, denom = ctx.pair_mash(dev, s)
        common, denom = common.cpu().numpy(), denom.cpu().numpy()
        for i in range(n):
            for j in range(0, n, 3):
                c, d = C.c_uint32(), C.c_uint32()
                L.orc_mash_counts_u64(sk[i].ctypes.data_as(C.c_void_p), len(sk[i]), sk[j].ctypes.data_as(C.c_void_p), len(sk[j]),
                                      s, C.byref(c), C.byref(d))
                assert (common[i, j], denom[i, j]) == (c.value, d.value), (s, i, j)
    # union of {1..6} and {2,4,..,12} in order: 1 2 3 4 5 6 | 8 10 12 ; the first six hold 3 shared values
    common, denom = ctx.pair_mash(dev, 6)
    assert (int(common[n - 2, n - 1]), int(denom[n - 2, n - 1])) == (3, 6)
    common, denom = ctx.pair_mash(dev, 1000)
    assert (int(common[n - 2, n - 1]), int(denom[n - 2, n - 1])) == (3, 9)
    assert int(common[4, 5]) == len(sk[4]) == int(denom[4, 5]) and int(denom[3, 3]) == 0
    dense = ctx.pair_common(dev).cpu().numpy()
    assert np.array_equal(common.cpu().numpy(), dense)  # with s beyond both lists the estimator counts every shared value
