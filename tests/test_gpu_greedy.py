"""GPU parity: greedy incremental clustering vs the oracle's -t 1 restatement of
MinHashGreedyClusterWithInvertedIndex / KssdGreedyClusterWithInvertedIndex (src/greedy.cpp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _family_sets(rng, n_fam, per, size, drift, pool_bits=40, dtype=np.uint64, ragged=False):
    """Families of sketches: members share most of the ancestor's hashes; exact duplicates included
    so that equal `common` ties occur."""
    out = []
    for f in range(n_fam):
        anc = np.unique(rng.integers(0, 1 << pool_bits, size=size * 2, dtype=np.uint64))[:size]
        for m in range(per):
            s = size if not ragged else int(size * rng.uniform(0.3, 1.0))
            keep = anc[: s].copy()
            if m % 3 != 0:  # every third member is an exact copy of a prefix of the ancestor
                nrep = int(len(keep) * drift * rng.uniform(0, 1))
                idx = rng.choice(len(keep), size=nrep, replace=False)
                keep[idx] = rng.integers(0, 1 << pool_bits, size=nrep, dtype=np.uint64)
            out.append(np.unique(keep).astype(dtype))
    order = rng.permutation(len(out))
    return [out[i] for i in order]


def test_greedy_fixed_size_fast_path(ctx, oracle):
    from rabbittclust_amd import api
    rng = np.random.default_rng(31)
    sk = _family_sets(rng, 60, 7, 400, 0.6)
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    flat, start, lens = oracle.to_csr(sk)
    for thr in (0.02, 0.05):
        want_n, want = oracle.greedy_minhash(flat, start, lens, 400, 21, False, thr)
        got_n, got = ctx.greedy(dev, thr, size_cfg=400, is_containment=False)
        assert got_n == want_n
        assert np.array_equal(got, want)


def test_greedy_containment_variable_sizes(ctx, oracle):
    from rabbittclust_amd import api
    rng = np.random.default_rng(32)
    sk = _family_sets(rng, 40, 6, 600, 0.5, ragged=True)
    cfg = np.array([max(len(s), 100) for s in sk], dtype=np.uint32)  # what getSketchSize() reports
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    flat, start, lens = oracle.to_csr(sk)
    want_n, want = oracle.greedy_minhash(flat, start, lens, cfg, 21, True, 0.05)
    got_n, got = ctx.greedy(dev, 0.05, size_cfg=cfg, is_containment=True)
    assert got_n == want_n
    assert np.array_equal(got, want)


def test_greedy_kssd_u32(ctx, oracle):
    from rabbittclust_amd import api
    rng = np.random.default_rng(33)
    sk = _family_sets(rng, 50, 6, 300, 0.5, pool_bits=30, dtype=np.uint32, ragged=True)
    sk.sort(key=lambda a: -len(a))  # caller-side std::sort by size, src/greedy.cpp:594-597
    dev = api.SketchSet.from_host(sk, ctx.device, k=22, kind="kssd", width=4)
    flat, start, lens = oracle.to_csr(sk, dtype=np.uint32)
    want_n, want = oracle.greedy_kssd(flat, start, lens, 22, 0.05)
    got_n, got = ctx.greedy(dev, 0.05)
    assert got_n == want_n
    assert np.array_equal(got, want)


def test_greedy_spans_several_batches(ctx, oracle):
    """More genomes than one 1024-query batch, so in-batch new representatives matter."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(34)
    sk = _family_sets(rng, 450, 6, 120, 0.7)
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    flat, start, lens = oracle.to_csr(sk)
    want_n, want = oracle.greedy_minhash(flat, start, lens, 120, 21, False, 0.05)
    got_n, got = ctx.greedy(dev, 0.05, size_cfg=120, is_containment=False)
    assert got_n == want_n
    assert np.array_equal(got, want)
    assert 1 < got_n < len(sk)


@pytest.mark.parametrize("mode", ["minhash", "containment", "kssd"])
def test_greedy_global_join_equals_the_batch_loop_and_the_oracle(ctx, oracle, mode, monkeypatch):
    """n > one batch: with the inverted join forced (RTC_PAIR_JOIN=2) rtc_greedy takes every co-occurring pair of the
    whole set from one join and replays; with it off (=0) it walks 1024-query batches against the representatives
    through the tiled kernel.  Same decisions either way, equal to the oracle's."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(35)
    if mode == "kssd":
        sk = _family_sets(rng, 420, 6, 150, 0.6, pool_bits=30, dtype=np.uint32, ragged=True)
        sk.sort(key=lambda a: -len(a))
        dev = api.SketchSet.from_host(sk, ctx.device, k=22, kind="kssd", width=4)
        flat, start, lens = oracle.to_csr(sk, dtype=np.uint32)
        want_n, want = oracle.greedy_kssd(flat, start, lens, 22, 0.05)
        run = lambda: ctx.greedy(dev, 0.05)
    else:
        cont = mode == "containment"
        sk = _family_sets(rng, 420, 6, 150, 0.6, ragged=cont)
        cfg = np.array([max(len(s), 100) for s in sk], dtype=np.uint32) if cont else 150
        dev = api.SketchSet.from_host(sk, ctx.device, k=21)
        flat, start, lens = oracle.to_csr(sk)
        want_n, want = oracle.greedy_minhash(flat, start, lens, cfg, 21, cont, 0.05)
        run = lambda: ctx.greedy(dev, 0.05, size_cfg=cfg, is_containment=cont)
    assert len(sk) > 2 * 1024
    # the library reads its switches when a context is created: ctx.env sets them AND makes this context read them again;
    # the path counters say which path really ran (a switch nobody read would leave the default path and a green test)
    for join in ("2", "0"):
        with ctx.env(RTC_PAIR_JOIN=join):
            d0 = ctx.diag()
            got_n, got = run()
            d1 = ctx.diag()
        assert got_n == want_n, join
        assert np.array_equal(got, want), join
        if join == "2":
            assert d1["greedy_global"] == d0["greedy_global"] + 1 and d1["greedy_blocks"] == d0["greedy_blocks"], (d0, d1)
        else:
            assert d1["greedy_global"] == d0["greedy_global"] and d1["greedy_blocks"] > d0["greedy_blocks"] and d1["join_tiles"] == d0["join_tiles"], (d0, d1)
    # the global join's pair list over its memory budget: the run falls through to the block loop, same decisions
    with ctx.env(RTC_PAIR_JOIN="2", RTC_GREEDY_GLOBAL_PAIRS="1000"):
        d0 = ctx.diag()
        got_n, got = run()
        d1 = ctx.diag()
    assert got_n == want_n and np.array_equal(got, want)
    assert d1["greedy_global"] == d0["greedy_global"] and d1["greedy_blocks"] > d0["greedy_blocks"], (d0, d1)
    assert 1 < want_n < len(sk)


import os
SOAK_SEEDS = int(os.environ.get("RTC_SOAK_SEEDS", "3"))  # RTC_SOAK_SEEDS=40: a longer walk


@pytest.mark.parametrize("seed", list(range(1, SOAK_SEEDS + 1)))
def test_greedy_on_random_families(ctx, oracle, seed):
    """Family structure, drift, sizes, raggedness and threshold drawn per seed; fixed-size MinHash, containment MinHash and
    KSSD u32 in turn: representative of every genome equal to the oracle's -t 1 replay."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(11000 + seed)
    n_fam, per = int(rng.integers(3, 60)), int(rng.integers(1, 9))
    size = int(rng.choice([50, 200, 500]))
    drift = float(rng.choice([0.1, 0.5, 0.9]))
    thr = float(rng.choice([0.02, 0.05, 0.1]))
    kind = seed % 3
    if kind == 0:
        sk = _family_sets(rng, n_fam, per, size, drift)
        sk = [s[:size] for s in sk if len(s) >= size] or [np.arange(size, dtype=np.uint64)]  # fixed-size mode: every sketch full
        dev = api.SketchSet.from_host(sk, ctx.device, k=21)
        flat, start, lens = oracle.to_csr(sk)
        want_n, want = oracle.greedy_minhash(flat, start, lens, size, 21, False, thr)
        got_n, got = ctx.greedy(dev, thr, size_cfg=size, is_containment=False)
    elif kind == 1:
        sk = _family_sets(rng, n_fam, per, size, drift, ragged=True)
        cfg = np.array([max(len(s), 100) for s in sk], dtype=np.uint32)
        dev = api.SketchSet.from_host(sk, ctx.device, k=21)
        flat, start, lens = oracle.to_csr(sk)
        want_n, want = oracle.greedy_minhash(flat, start, lens, cfg, 21, True, thr)
        got_n, got = ctx.greedy(dev, thr, size_cfg=cfg, is_containment=True)
    else:
        sk = _family_sets(rng, n_fam, per, size, drift, pool_bits=30, dtype=np.uint32, ragged=True)
        sk.sort(key=lambda a: -len(a))
        dev = api.SketchSet.from_host(sk, ctx.device, k=22, kind="kssd", width=4)
        flat, start, lens = oracle.to_csr(sk, dtype=np.uint32)
        want_n, want = oracle.greedy_kssd(flat, start, lens, 22, thr)
        got_n, got = ctx.greedy(dev, thr)
    assert got_n == want_n, (seed, kind)
    assert np.array_equal(got, want), (seed, kind)
