"""GPU checks at BASELINE.json's full single-GPU size (10 000 x 5 Mbp, k=21, s=1000) through
size-independent properties, plus oracle spot checks on a sample of genomes and pairs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(ctx):
    from rabbittclust_amd import api
    free, _ = torch.cuda.mem_get_info()
    n, L = 10000, 5_000_000
    if free < 80e9:
        pytest.skip("needs ~60 GB of HBM")
    desc = api.synth_family_descs(n // 10, 10, global_seed=42)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sk = ctx.sketch_minhash(seq, off, k=21, size=1000)
    ctx.sync()
    yield dict(n=n, L=L, desc=desc, off=off, seq=seq, sk=sk)
    del seq


def test_full_size_sketch_properties_and_oracle_sample(ctx, oracle, full):
    sk, n, L = full["sk"], full["n"], full["L"]
    h = sk.hashes.view(n, -1)
    assert int(sk.len.min()) == 1000 == int(sk.len.max())
    # strictly ascending rows (sorted + distinct), compared as unsigned
    hu = (h ^ torch.tensor(-2 ** 63, dtype=torch.int64, device=h.device))  # order-preserving u64 -> i64 map
    assert bool((hu[:, 1:] > hu[:, :-1]).all())
    # idempotence: sketching again gives the same bits
    again = ctx.sketch_minhash(full["seq"], full["off"], k=21, size=1000)
    assert torch.equal(again.hashes, sk.hashes)
    # oracle on a sample of genomes (first, a mutated family member, last)
    for g in (0, 4321, n - 1):
        d = full["desc"][g]
        ref_seq = oracle.synth_genome(int(d["fam_seed"]), int(d["mut_seed"]), int(d["mut_thr"]), L)
        dev_seq = full["seq"][g * L:(g + 1) * L].cpu().numpy()
        assert np.array_equal(ref_seq, dev_seq)
        want = oracle.sketch_minhash_batch(ref_seq, np.array([0, L], dtype=np.uint64), 21, 1000)[0]
        assert np.array_equal(h[g].cpu().numpy().view(np.uint64), want), g


def test_full_size_pairs_mst_properties(ctx, oracle, full):
    from rabbittclust_amd import api, pipeline
    sk, n = full["sk"], full["n"]
    # a 512 x 512 diagonal block: symmetric, diagonal = |A|, and equal to the merge kernel
    a = ctx.pair_common(sk, row0=4096, row1=4608, col0=4096, col1=4608, algo=2)
    b = ctx.pair_common(sk, row0=4096, row1=4608, col0=4096, col1=4608, algo=1)
    assert torch.equal(a, b) and torch.equal(a, a.t())
    assert bool((a.diagonal() == 1000).all())
    # an off-diagonal block against the oracle on a few pairs
    blk = ctx.pair_common(sk, row0=9000, row1=9064, col0=0, col1=9000, algo=0).cpu().numpy()
    host = {g: sk.hashes.view(n, -1)[g].cpu().numpy().view(np.uint64) for g in (9000, 9005, 9063, 0, 17, 8999, 9001)}
    for r in (9000, 9005, 9063):
        for c in (0, 17, 8999):
            assert blk[r - 9000, c] == oracle.common(host[r], host[c])
    # whole pipeline: forest size = n - components; every family ends up in one component at d=0.1
    pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=1000, threshold=0.05)
    edges, m = pipe.candidate_edges(sk, 0, n)
    sel, rounds = pipe.boruvka(sk, edges, m)
    mst = pipe.finish(sk, sel)
    parent = np.arange(n)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for e in mst:
        ra, rb = find(int(e["preNode"])), find(int(e["sufNode"]))
        assert ra != rb  # a forest: no edge closes a cycle
        parent[ra] = rb
    comps = len({find(v) for v in range(n)})
    assert len(mst) == n - comps
    assert bool(np.all(np.diff(mst["dist"]) >= 0))  # sorted by distance
    # members of one family (substitution rate <= 8 %) are linked below d = 0.1
    close = mst[mst["dist"] <= 0.1]
    parent = np.arange(n)
    for e in close:
        parent[find(int(e["preNode"]))] = find(int(e["sufNode"]))
    for f in (0, 123, 999):
        assert len({find(f * 10 + m_) for m_ in range(10)}) == 1
    # oracle MST on the first 300 genomes equals ours restricted to them (weights multiset)
    sub = api.SketchSet(sk.hashes[: 300 * 1000], sk.start[:300], sk.len[:300], 8, 21, "minhash")
    got = ctx.mst(sub, 0.05)
    flat, start, lens = oracle.to_csr(sub.to_host())
    want = oracle.mst(flat, start, lens, 21, 0, 0.05)
    assert np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))


# ---- BASELINE config 4: clust-greedy, containment (-c 1000), 50 000 genomes of 0.4 .. 2 Mbp ----------
def test_config4_greedy_containment_50k(ctx, oracle):
    """Full rtc_greedy on 50 000 variable-size sketches (prefix genomes: families of 10 where member m
    is a random-length prefix of the ancestor, sketch size = ~bytes/1000 like -c 1000).  Invariants at
    full size; representative assignment identical to the oracle's greedy over ALL 50 000 sketches (the rows come down
    in one copy; the oracle's greedy is an index walk, seconds at this size)."""
    from rabbittclust_amd import api
    free, _ = torch.cuda.mem_get_info()
    if free < 90e9:
        pytest.skip("needs ~70 GB of HBM")
    n, L, fam = 50000, 2_000_000, 10
    rng = np.random.default_rng(1)
    desc = api.synth_family_descs(n // fam, fam, global_seed=43, max_rate=0.04)
    for f in range(n // fam):  # members share the ancestor's mutation stream too: true prefixes of one genome
        desc[f * fam:(f + 1) * fam] = desc[f * fam]
    frac = rng.uniform(0.2, 1.0, size=n)
    frac[::fam] = 1.0
    lens = (frac * L).astype(np.uint64) // 16 * 16
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    seq = ctx.synth_genomes(desc, off)
    sizes = np.maximum((lens.astype(np.float64) * 1.0125 / 1000).astype(np.uint32), 100)  # max(fileBytes / 1000, 100)
    sk = ctx.sketch_minhash(seq, off, k=21, sizes=sizes)
    ctx.sync()
    lens_sk = sk.len.cpu().numpy()
    assert np.array_equal(lens_sk, sizes) and int(sizes.max()) > 2000 and int(sizes.min()) < 450
    ncl, rep = ctx.greedy(sk, 0.05, size_cfg=sizes, is_containment=True)
    # invariants: representatives represent themselves, come earlier, clusters stay inside families
    idx = np.arange(n)
    assert np.all(rep[rep] == rep) and np.all(rep <= idx) and ncl == int((rep == idx).sum())
    assert np.all(rep // fam == idx // fam), "a genome joined a representative of another family"
    assert n // fam <= ncl < n // 2
    # members are within the threshold of their representative (greedy's distance, oracle arithmetic)
    stride = sk.hashes.numel() // n
    rows = sk.hashes.view(n, stride)
    for g in rng.choice(np.nonzero(rep != idx)[0], size=60, replace=False):
        r = int(rep[g])
        a = rows[g, :int(sizes[g])].cpu().numpy().view(np.uint64)
        b = rows[r, :int(sizes[r])].cpu().numpy().view(np.uint64)
        c = oracle.common(a, b)
        d = oracle.lib().orc_greedy_distance(c, int(sizes[g]), int(sizes[r]), 21, 1)
        assert d <= 0.05, (g, r, c, d)
    # the oracle on the whole set: one bulk copy of the rows
    m = n
    rows_h = rows.cpu().numpy().view(np.uint64)
    host = [rows_h[g, :int(sizes[g])] for g in range(m)]
    # spot-check the sketches themselves against the oracle sketcher
    for g in (0, 7, 1234):
        ref = oracle.synth_genome(int(desc[g]["fam_seed"]), int(desc[g]["mut_seed"]), int(desc[g]["mut_thr"]), int(lens[g]))
        assert np.array_equal(oracle.sketch_minhash_batch(ref, np.array([0, lens[g]], dtype=np.uint64), 21, int(sizes[g]))[0], host[g])
    flat, start, ln = oracle.to_csr(host)
    want_n, want = oracle.greedy_minhash(flat, start, ln, sizes[:m], 21, True, 0.05)
    assert np.array_equal(rep[:m], want) and want_n == int((rep[:m] == idx[:m]).sum())
    del seq


# ---- BASELINE config 5 per-GPU shape: --fast (KSSD), 25 000 x 2 Mbp --------------------------------
def test_config5_kssd_per_gpu_shape_25k(ctx, oracle):
    from rabbittclust_amd import api, host, pipeline
    free, _ = torch.cuda.mem_get_info()
    if free < 80e9:
        pytest.skip("needs ~55 GB of HBM")
    n, L = 25000, 2_000_000
    desc = api.synth_family_descs(n // 10, 10, global_seed=45)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sd = host.generate_shuffle_dim(6)
    sk = ctx.sketch_kssd(seq, off, sd, kmer_size=21, drlevel=3)
    ctx.sync()
    assert sk.width == 4 and sk.k == 22
    stride = sk.hashes.numel() // n
    rows = sk.hashes.view(n, stride)
    ln = sk.len.to(torch.int64)
    assert 380 < float(ln.float().mean()) < 600  # ~L / 4096
    # rows ascending + distinct over their length (u32 compared through an order-preserving map)
    u = rows.to(torch.int64) & 0xFFFFFFFF
    pos = torch.arange(stride, device=rows.device)[None, :]
    ok = (u[:, 1:] > u[:, :-1]) | (pos[:, 1:] >= ln[:, None])
    assert bool(ok.all())
    again = ctx.sketch_kssd(seq, off, sd, kmer_size=21, drlevel=3)
    assert torch.equal(again.len, sk.len)
    st2 = again.hashes.numel() // n
    valid = pos < ln[:, None]
    assert torch.equal(torch.where(valid, rows, 0), torch.where(valid[:, :st2] if st2 >= stride else valid, again.hashes.view(n, st2)[:, :stride], 0))
    for g in (0, 11111, n - 1):
        ref = oracle.synth_genome(int(desc[g]["fam_seed"]), int(desc[g]["mut_seed"]), int(desc[g]["mut_thr"]), L)
        assert np.array_equal(rows[g, :int(ln[g])].cpu().numpy().view(np.uint32), oracle.kssd_sketch(ref, 21, 3)), g
    # whole step at this shape: forest properties + oracle weights on a 300-genome prefix
    pipe = pipeline.MstPipeline(ctx, k=21, threshold=0.05, mode="kssd", shuffled_dim=sd)
    st = pipe.step(seq, off)
    mst = pipe.last_mst
    assert st["mst_edges"] == len(mst) and bool(np.all(np.diff(mst["dist"]) >= 0))
    parent = np.arange(n)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for e in mst:
        ra, rb = find(int(e["preNode"])), find(int(e["sufNode"]))
        assert ra != rb
        parent[ra] = rb
    sub = api.SketchSet(sk.hashes[: 300 * stride], sk.start[:300], sk.len[:300], 4, 22, "kssd")
    got = ctx.mst(sub, 0.05)
    flat, start, lens = oracle.to_csr(sub.to_host(), dtype=np.uint32)
    want = oracle.mst(flat, start, lens, 22, 0, 0.05)
    assert len(got) == len(want) and np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
    del seq


# ---- BASELINE configs 3 and 5 at their real pair-space size: 8 ranks of rtc_mst_sharded on one GPU ----------
def _sharded_in_process(world, sk, threshold, width_k_kind=None):
    """`world` contexts on device 0, one host thread each, rtc_comm_init_all's in-process exchange: every rank runs
    rtc_mst_sharded (its triangle row range with the product's own split, one reduction per Boruvka round in
    fixed-size mode, three otherwise).  Returns the per-rank (edge.mst records, ShardStats)."""
    from rabbittclust_amd import api
    from test_gpu_multigpu import _threads
    ctxs = [api.Context(0) for _ in range(world)]
    comms = api.Comm.init_all(ctxs)
    assert all(c.backend == "in-process" for c in comms)
    torch.cuda.synchronize()
    try:
        return _threads([(lambda r=r: ctxs[r].mst_sharded(comms[r], sk, threshold)) for r in range(world)])
    finally:
        for c in comms:
            c.close()
        for c in ctxs:
            c.close()


def _check_forest(mst, n):
    parent = np.arange(n)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for e in mst:
        ra, rb = find(int(e["preNode"])), find(int(e["sufNode"]))
        assert ra != rb  # a forest: no edge closes a cycle
        parent[ra] = rb
    assert bool(np.all(np.diff(mst["dist"]) >= 0))
    return find


def test_config3_100k_sketches_8_ranks_full_pair_space(ctx, oracle):
    """BASELINE config 3 at its real size: 100 000 x 5 Mbp genomes (500 Gbp, synthesised and sketched in ten 50 GB chunks
    of 10 000 genomes; k=21, s=1000) and the 5*10^9-pair space over their sketches.  All 8 triangle row ranges go through rtc_mst_sharded
    itself (8 in-process ranks, fixed-size mode: ONE reduction per Boruvka round) and every rank must return the
    forest of the single rtc_mst launch over the 5*10^9 pairs, bit for bit; rank 7's candidate list is checked against
    the independent merge kernel and the oracle on sampled rows; a 3 000-sketch prefix against the oracle's MST."""
    from rabbittclust_amd import api, pipeline
    free, _ = torch.cuda.mem_get_info()
    if free < 80e9:
        pytest.skip("needs ~60 GB of HBM")
    n, L, chunk, s, world = 100000, 5_000_000, 10000, 1000, 8
    out = torch.empty((n, s), dtype=torch.int64, device=ctx.device)
    cnt = torch.zeros(n, dtype=torch.int32, device=ctx.device)
    off = np.arange(chunk + 1, dtype=np.uint64) * np.uint64(L)
    for c0 in range(0, n, chunk):  # one 50 GB staging buffer
        desc = api.synth_family_descs(chunk // 10, 10, global_seed=500 + c0)
        seq = ctx.synth_genomes(desc, off)
        ctx.sketch_minhash_into(seq, off, out[c0:c0 + chunk], cnt[c0:c0 + chunk], k=21, size=s)
        ctx.sync()
        del seq
    sk = api.SketchSet(out.view(-1), torch.arange(n, dtype=torch.int64, device=ctx.device) * s, cnt, 8, 21, "minhash")
    assert int(cnt.min()) == s == int(cnt.max())
    single = ctx.mst(sk, 0.05)
    find = _check_forest(single, n)
    for f in (0, 4321, 9999):  # families (substitution rate <= 8 %) hang together in the forest
        assert len({find(f * 10 + m_) for m_ in range(10)}) == 1
    res = _sharded_in_process(world, sk, 0.05)
    bounds = pipeline.triangle_row_ranges(n, world, fixed_cols=1.84 * s)  # the product's split (rtc_mst_sharded)
    tot_edges = 0
    for r, (mst, st) in enumerate(res):
        assert (int(st.row0), int(st.row1)) == (bounds[r], bounds[r + 1]) and int(st.s_fixed) == s and int(st.contractions) == 0
        assert np.array_equal(mst, single), f"rank {r} ended with a different forest"
        tot_edges += int(st.cand_edges)
    assert all(int(st.rounds) == int(res[0][1].rounds) for _, st in res) and int(res[0][1].rounds) >= 3
    # rank 7's candidate list against the merge kernel and the oracle on sampled rows
    pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=s, threshold=0.05)
    e7, m7 = pipe.candidate_edges(sk, bounds[7], bounds[8])
    assert m7 == int(res[7][1].cand_edges)
    edges7 = e7[:m7].cpu().numpy().view(np.uint32)
    assert np.all(edges7[:, 1] < edges7[:, 0]) and np.all(edges7[:, 0] >= bounds[7]) and np.all(edges7[:, 2] > 0)
    assert len(np.unique(edges7[:, 0].astype(np.uint64) << np.uint64(32) | edges7[:, 1])) == m7
    rng = np.random.default_rng(3)
    for row in rng.integers(bounds[7], n, size=5):
        dense = ctx.pair_common(sk, row0=int(row), row1=int(row) + 1, col0=0, col1=int(row), algo=1).cpu().numpy()[0]
        mine = edges7[edges7[:, 0] == row]
        want_cols = np.nonzero(dense)[0]
        assert np.array_equal(np.sort(mine[:, 1]), want_cols) and np.array_equal(mine[np.argsort(mine[:, 1]), 2], dense[want_cols])
        for col in want_cols[:3]:
            assert oracle.common(out[row].cpu().numpy().view(np.uint64), out[int(col)].cpu().numpy().view(np.uint64)) == dense[col]
    # total candidates of the 8 ranges == the single launch's list
    e_all, m_all = pipe.candidate_edges(sk, 0, n)
    assert m_all == tot_edges
    # a 3 000-sketch prefix through the same 8-rank code against the oracle
    m = 3000
    sub = api.SketchSet(out[:m].reshape(-1), sk.start[:m], cnt[:m], 8, 21, "minhash")
    res2 = _sharded_in_process(world, sub, 0.05)
    flat, start, lens = oracle.to_csr(sub.to_host())
    want = oracle.mst(flat, start, lens, 21, 0, 0.05, threads=8)
    for mst, _ in res2:
        assert np.array_equal(np.sort(mst["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
        assert np.array_equal(mst, res2[0][0])


def test_config5_200k_kssd_sketches_8_ranks_full_pair_space(ctx, oracle):
    """BASELINE config 5 at its real size: 200 000 x 2 Mbp genomes, --fast (KSSD k=21 -> 22, drlevel 3, u32 tuples of
    variable count), sketched as 8 chunks of 25 000 (what each of the 8 GPUs would sketch).  The 2*10^10-pair space
    goes through rtc_mst_sharded on 8 in-process ranks (variable sizes: the three-reduction Boruvka round) and every
    rank must return the forest of the single rtc_mst launch; oracle on sampled sketches and on a 3 000-sketch prefix."""
    from rabbittclust_amd import api, host
    free, _ = torch.cuda.mem_get_info()
    if free < 90e9:
        pytest.skip("needs ~60 GB of HBM")
    n, L, chunk, world = 200000, 2_000_000, 25000, 8
    sd = host.generate_shuffle_dim(6)
    stride = L // 4096 * 3 // 2 + 256
    rows = torch.zeros((n, stride), dtype=torch.int32, device=ctx.device)
    ln = torch.zeros(n, dtype=torch.int32, device=ctx.device)
    off = np.arange(chunk + 1, dtype=np.uint64) * np.uint64(L)
    descs = {}
    for c0 in range(0, n, chunk):  # one 50 GB staging buffer
        desc = api.synth_family_descs(chunk // 10, 10, global_seed=900 + c0)
        seq = ctx.synth_genomes(desc, off)
        part = ctx.sketch_kssd(seq, off, sd, kmer_size=21, drlevel=3, stride=stride)
        ctx.sync()
        pst = part.hashes.numel() // chunk
        assert part.width == 4 and part.k == 22 and pst == stride
        rows[c0:c0 + chunk] = part.hashes.view(chunk, pst)
        ln[c0:c0 + chunk] = part.len
        for g in (c0, c0 + chunk - 1):  # the sketches themselves against the oracle sketcher
            d = desc[g - c0]
            ref = oracle.synth_genome(int(d["fam_seed"]), int(d["mut_seed"]), int(d["mut_thr"]), L)
            assert np.array_equal(part.hashes.view(chunk, pst)[g - c0, :int(part.len[g - c0])].cpu().numpy().view(np.uint32),
                                  oracle.kssd_sketch(ref, 21, 3)), g
        del seq, part
    torch.cuda.empty_cache()
    sk = api.SketchSet(rows.view(-1), torch.arange(n, dtype=torch.int64, device=ctx.device) * stride, ln, 4, 22, "kssd")
    assert 380 < float(ln.float().mean()) < 600 and int(ln.min()) != int(ln.max())
    single = ctx.mst(sk, 0.05)
    find = _check_forest(single, n)
    for f in (0, 7777, 19999):
        assert len({find(f * 10 + m_) for m_ in range(10)}) == 1
    res = _sharded_in_process(world, sk, 0.05)
    for r, (mst, st) in enumerate(res):
        assert int(st.s_fixed) == 0 and int(st.row1) > int(st.row0)
        assert np.array_equal(mst, single), f"rank {r} ended with a different forest"
    assert [int(st.row0) for _, st in res][1:] == [int(st.row1) for _, st in res][:-1] and int(res[-1][1].row1) == n
    # prefix against the oracle (u32 index-based compute_kssd_mst restatement)
    m = 3000
    sub = api.SketchSet(rows[:m].reshape(-1), sk.start[:m], ln[:m], 4, 22, "kssd")
    res2 = _sharded_in_process(world, sub, 0.05)
    flat, start, lens = oracle.to_csr(sub.to_host(), dtype=np.uint32)
    want = oracle.mst(flat, start, lens, 22, 0, 0.05, threads=8)
    for mst, _ in res2:
        assert len(mst) == len(want) and np.array_equal(np.sort(mst["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))


def test_chromosome_sized_genomes_segments_and_passes(ctx, oracle):
    """Maximum sizes: two genomes of 100 and 200 Mbp in one batch (768 partial segments that start from the genomes'
    thresholds, merged on the device) at s = 1 000, and 40 + 80 Mbp at s = 20 000 (four passes over ascending hash
    ranges, the later ones without a threshold); an N run inside.  Bit-identical to the oracle."""
    for L, s in ((300_000_000, 1000), (120_000_000, 20000)):
        g = oracle.synth_genome(99, 7, 100, L)
        g[10_000_000:10_000_500] = ord("N")
        off = np.array([0, L // 3, L], dtype=np.uint64)
        d = ctx.upload_sequences(g)
        sk = ctx.sketch_minhash(d, off, k=21, size=s)
        ctx.sync()
        got = sk.to_host()
        want = oracle.sketch_minhash_batch(g, off, 21, s)
        assert [len(x) for x in got] == [s, s]
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), (L, s)
        del d, sk
    torch.cuda.empty_cache()
