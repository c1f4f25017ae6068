"""GPU parity in the DENSE regime of the pair phase: families of hundreds of near-identical genomes (posting lists as long
as a family -- where the reference's posting-list walk, src/MST.cpp:1412-1435, goes quadratic).  The cost rule of
rtc_pair_edges_dev has to send these inputs to the tiled N x N kernel BY ITSELF (no RTC_PAIR_JOIN override), and what
comes back must equal the oracle's inverted-index counts pair for pair, and the oracle's MST."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_edges(oracle, sk_host, radio):
    flat, start, lens = oracle.to_csr(sk_host, dtype=sk_host[0].dtype)
    cp = oracle.candidate_pairs(flat, start, lens)            # every pair sharing a hash: (i, j, common)
    a = np.stack([cp["pre"], cp["suf"], cp["common"]], axis=1).astype(np.int64)
    lo, hi = np.minimum(a[:, 0], a[:, 1]), np.maximum(a[:, 0], a[:, 1])
    a[:, 0], a[:, 1] = hi, lo                                 # rows are the larger index (j < i)
    li, lj = lens[a[:, 0]].astype(np.int64), lens[a[:, 1]].astype(np.int64)
    keep = np.maximum(li, lj) <= radio * np.minimum(li, lj)  # src/MST.cpp:1481-1484
    a = a[keep]
    return a[np.lexsort((a[:, 1], a[:, 0]))]


def _gpu_edges(ctx, sk, radio):
    assert "RTC_PAIR_JOIN" not in os.environ, "this test is about the default dispatch"
    cap = sk.n * (sk.n - 1) // 2 + 16
    e, m = ctx.pair_edges(sk, 1, sk.n, 0, sk.n - 1, radio, cap)
    assert m <= cap
    a = e[:m].cpu().numpy().view(np.uint32).astype(np.int64)
    return a[np.lexsort((a[:, 1], a[:, 0]))]


def test_dense_minhash_families_default_dispatch_equals_oracle(ctx, oracle):
    from rabbittclust_amd import api
    n_fam, per, L = 3, 500, 200_000
    desc = api.synth_family_descs(n_fam, per, global_seed=11, max_rate=0.01)
    off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sk = ctx.sketch_minhash(seq, off, k=21, size=1000)
    host = sk.to_host()
    radio = api.mst_radio(0.05, 21)
    got = _gpu_edges(ctx, sk, radio)
    assert ctx.pair_last_path() == 2, "the cost rule must pick the tiled kernel on a dense input"
    assert ctx.pair_last_kernel_ms() > 0
    want = _oracle_edges(oracle, host, radio)
    assert len(want) >= n_fam * per * (per - 1) // 2           # every pair of a family is a candidate
    assert np.array_equal(got, want)
    # a second launch over the same sketches (the memo of the refusal is keyed on buffer + sketch generation): same triples
    assert np.array_equal(_gpu_edges(ctx, sk, radio), want)
    assert ctx.pair_last_path() == 2
    # and the MST of the flow that sits on top of it
    flat, start, lens = oracle.to_csr(host)
    ref = oracle.mst(flat, start, lens, 21, 0, 0.05)
    mst = ctx.mst(sk, 0.05)
    assert len(mst) == len(ref) == n_fam * per - n_fam
    assert np.array_equal(np.sort(mst["dist"]).view(np.uint64), np.sort(ref["dist"]).view(np.uint64))


@pytest.mark.parametrize("width", [4, 8])
def test_dense_built_families_ragged_sizes(ctx, oracle, width):
    """Hand-built dense sets: 4 families of 300 sketches drawn from a family pool (sizes 200 .. 1000, so the size-ratio
    filter bites), one family whose members are IDENTICAL, plus empty sketches; u32 (KSSD width) and u64."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(5 + width)
    dt = np.uint32 if width == 4 else np.uint64
    top = np.iinfo(dt).max
    sk = []
    for f in range(4):
        pool = np.unique(rng.integers(0, top, size=1300, dtype=np.uint64).astype(dt))
        for m in range(300):
            if f == 3:
                sk.append(pool[:700].copy())
                continue
            s = int(rng.integers(200, 1001))
            sk.append(np.sort(rng.choice(pool, size=s, replace=False)).astype(dt))
    sk[17] = np.zeros(0, dtype=dt)
    sk[900] = np.zeros(0, dtype=dt)
    sk[5] = np.concatenate([sk[5][sk[5] != top], np.array([top], dtype=dt)])   # the largest value of the type as a hash
    sk[6] = np.concatenate([sk[6][sk[6] != top], np.array([top], dtype=dt)])
    dev = api.SketchSet.from_host(sk, ctx.device, width=width)
    for radio in (4, 2):
        got = _gpu_edges(ctx, dev, radio)
        assert ctx.pair_last_path() == 2
        assert np.array_equal(got, _oracle_edges(oracle, sk, radio)), radio
