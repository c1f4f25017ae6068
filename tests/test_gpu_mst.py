"""GPU parity: candidate edges and the minimum spanning forest vs the oracle's restatement of
compute_minhash_mst (src/MST.cpp:1290-1737).  Ties make the edge SET ambiguous (the reference
uses an unstable sort), so the asserted invariants are: identical sorted multiset of edge
weights (bit-for-bit doubles), identical partition at the threshold, forest size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reload_options():
    """the library reads its RTC_* switches when a context is created: every live context reads them again"""
    from rabbittclust_amd import api
    api.reload_all_options()


def _partition(clusters):
    return sorted(tuple(sorted(c)) for c in clusters)


def _clusters_from_edges(edges, thr, n):
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for e in edges:
        if e["dist"] <= thr:
            a, b = find(int(e["preNode"])), find(int(e["sufNode"]))
            if a != b:
                parent[a] = b
    groups = {}
    for v in range(n):
        groups.setdefault(find(v), []).append(v)
    return list(groups.values())


def _family_sketches(ctx, n_fam, per, L, size=1000, seed=3):
    from rabbittclust_amd import api
    desc = api.synth_family_descs(n_fam, per, global_seed=seed)
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    return ctx.sketch_minhash(seq, off, k=21, size=size), seq, off


@pytest.mark.parametrize("containment", [False, True])
def test_mst_matches_oracle_on_families(ctx, oracle, containment):
    sk, _, _ = _family_sketches(ctx, 12, 6, 100_000)
    host = sk.to_host()
    flat, start, lens = oracle.to_csr(host)
    for thr in (0.01, 0.05, 0.1):
        got = ctx.mst(sk, thr, is_containment=containment)
        want = oracle.mst(flat, start, lens, 21, containment, thr, threads=1)
        assert len(got) == len(want)
        assert np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
        pw = _partition(oracle.forest_clusters(want, thr, len(host)))
        pg = _partition(_clusters_from_edges(got, thr, len(host)))
        assert pw == pg


def test_mst_random_overlapping_sets_many_ties(ctx, oracle):
    from rabbittclust_amd import api
    rng = np.random.default_rng(5)
    pool = np.unique(rng.integers(1, 1 << 62, size=4000, dtype=np.uint64))
    sk = []
    for g in range(200):
        size = int(rng.integers(20, 200))
        sk.append(np.sort(rng.choice(pool, size=size, replace=False)))
    sk[17] = np.zeros(0, dtype=np.uint64)
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    flat, start, lens = oracle.to_csr(sk)
    for containment in (False, True):
        got = ctx.mst(dev, 0.05, is_containment=containment)
        want = oracle.mst(flat, start, lens, 21, containment, 0.05, threads=1)
        assert len(got) == len(want)
        assert np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
        for thr in (0.05, 0.2, 0.4):
            assert _partition(oracle.forest_clusters(want, thr, 200)) == _partition(_clusters_from_edges(got, thr, 200))


def test_extract_edges_equals_oracle_candidates(ctx, oracle):
    from rabbittclust_amd import api
    sk, _, _ = _family_sketches(ctx, 8, 5, 60_000, seed=9)
    host = sk.to_host()
    n = len(host)
    flat, start, lens = oracle.to_csr(host)
    cand = oracle.candidate_pairs(flat, start, lens)
    common = ctx.pair_common(sk, lower_only=True)
    radio = api.mst_radio(0.05, 21)
    edges, count = ctx.extract_edges(common, sk, 0, n, 0, n, radio, cap=n * n)
    m = int(count.item())
    got = edges[:m].cpu().numpy().view(np.uint32)
    got = sorted(map(tuple, got.tolist()))
    want = sorted((int(e["pre"]), int(e["suf"]), int(e["common"])) for e in cand)  # equal sizes -> radio passes
    assert got == want
    # the fused form (pair kernel emits the edges itself) gives the same list
    fused, m2 = ctx.pair_edges(sk, 0, n, 0, n, radio, cap=n * n)
    assert sorted(map(tuple, fused[:m2].cpu().numpy().view(np.uint32).tolist())) == want
    # a capacity that is too small: everything is still counted, nothing is written past the end
    small, m3 = ctx.pair_edges(sk, 0, n, 0, n, radio, cap=7)
    assert m3 == len(want) and set(map(tuple, small[:7].cpu().numpy().view(np.uint32).tolist())) <= set(want)
    # row sub-range, merge-path fallback
    import os
    r0, r1 = 11, 29
    sub = sorted(t for t in want if r0 <= t[0] < r1)
    e1, m4 = ctx.pair_edges(sk, r0, r1, 0, r1 - 1, radio, cap=n * n)
    assert sorted(map(tuple, e1[:m4].cpu().numpy().view(np.uint32).tolist())) == sub
    os.environ["RTC_PAIR_FORCE_MERGE"] = "1"
    _reload_options()
    try:
        e2, m5 = ctx.pair_edges(sk, r0, r1, 0, r1 - 1, radio, cap=n * n)
    finally:
        del os.environ["RTC_PAIR_FORCE_MERGE"]
        _reload_options()
    assert sorted(map(tuple, e2[:m5].cpu().numpy().view(np.uint32).tolist())) == sub


def test_fused_edges_dense_stress_and_radio(ctx, oracle):
    """Worst case for the edge emission: every pair of 1 500 mutually similar sketches is an edge
    (1.1 M edges from 24 x 2 workgroups, every lane writes), plus variable sizes so that the radio
    test (src/MST.cpp:1481-1484) actually rejects pairs."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(11)
    core = np.unique(rng.integers(1, 1 << 62, size=400, dtype=np.uint64))
    pool = np.unique(rng.integers(1, 1 << 62, size=6000, dtype=np.uint64))
    sk = []
    for g in range(1500):
        extra = rng.choice(pool, size=int(rng.integers(0, 1400)), replace=False)
        sk.append(np.unique(np.concatenate([core[: int(rng.integers(50, 400))], extra])))
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    n = len(sk)
    lens = np.array([len(x) for x in sk])
    radio = api.mst_radio(0.05, 21)
    dense = ctx.pair_common(dev, lower_only=True).cpu().numpy()
    want = []
    for i in range(n):
        for j in range(i):
            c = int(dense[i, j])
            if c > 0 and max(lens[i], lens[j]) <= radio * min(lens[i], lens[j]):
                want.append((i, j, c))
    assert len(want) > 500_000 and len(want) < n * (n - 1) // 2  # dense, and the radio test rejected some
    # spot check of the dense counts against the oracle
    for (i, j) in ((5, 2), (700, 3), (1499, 1498), (900, 450)):
        assert dense[i, j] == oracle.common(sk[i], sk[j])
    for trial in range(3):  # atomics order varies run to run; the set must not
        e, m = ctx.pair_edges(dev, 0, n, 0, n, radio, cap=n * n // 2)
        assert m == len(want)
        got = e[:m].cpu().numpy().view(np.uint32)
        assert sorted(map(tuple, got.tolist())) == want
    e, m = ctx.pair_edges(dev, 0, n, 0, n, -1, cap=n * n // 2)  # radio < 0: no size test (greedy)
    assert m == int((np.tril(dense, -1) > 0).sum())


def test_mst_dense_input_edge_budget_contraction(ctx, oracle):
    """Dense inputs beyond the edge budget: rows are walked in chunks and the list is contracted to
    its own spanning forest in between (RTC_EDGE_BUDGET shrinks the budget so 600 sketches trigger it);
    the forest must equal the unbudgeted one."""
    import os
    from rabbittclust_amd import api
    rng = np.random.default_rng(13)
    core = np.unique(rng.integers(1, 1 << 62, size=300, dtype=np.uint64))
    pool = np.unique(rng.integers(1, 1 << 62, size=3000, dtype=np.uint64))
    sk = [np.unique(np.concatenate([core[: int(rng.integers(100, 300))], rng.choice(pool, size=200, replace=False)]))
          for _ in range(600)]
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    flat, start, lens = oracle.to_csr(sk)
    want = oracle.mst(flat, start, lens, 21, 0, 0.05, threads=1)
    ref = ctx.mst(dev, 0.05)
    os.environ["RTC_EDGE_BUDGET"] = "40000"
    _reload_options()
    try:
        got = ctx.mst(dev, 0.05)
    finally:
        del os.environ["RTC_EDGE_BUDGET"]
        _reload_options()
    assert np.array_equal(got, ref)
    assert np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))


def test_mst_when_the_join_estimate_of_the_edge_count_is_short(oracle):
    """One family of 8 000 sketches (s = 500) at pairwise Jaccard ~0.3, default dispatch: the sort would be cheaper than the tiled
    kernel, so the join looks at its density sample, refuses the set and hands back an ESTIMATE of the candidate edges (1.5 x n x
    (g - 1) / 2 with g = the posting-list length a sampled hash sees, ~0.46 n: 22 M) that is SHORT of the truth -- every one of
    the 32 M pairs shares a hash.  The list grown to the estimate is too short again; rtc_candidate_edges_device must grow it a
    second time from the exact count and never hand Boruvka a count past the allocation (round 5's advisor finding: el->m = cnt
    > cap).  Checked against the forest of the tiled kernel alone (RTC_PAIR_JOIN=0: no sample, no estimate) and, edge by edge,
    against intersections counted in numpy."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(77)
    n, s = 8000, 500
    pool = np.unique(rng.integers(1, 1 << 62, size=1500, dtype=np.uint64))[:1090]
    member = np.zeros((n, len(pool)), dtype=np.float32)
    sk = []
    for g in range(n):
        idx = np.sort(rng.choice(len(pool), size=s, replace=False))
        member[g, idx] = 1.0
        sk.append(pool[idx])
    ctx = api.Context(0)  # a context of its own: the session's may hold a long edge list from an earlier clustering call (it keeps the last one)
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    d0 = ctx.diag()
    got = ctx.mst(dev, 0.05)
    d1 = ctx.diag()
    assert d1["estimates"] == d0["estimates"] + 1, (d0, d1)                       # the early-out was taken ...
    assert d1["tiled_tiles"] >= d0["tiled_tiles"] + 2, (d0, d1)                   # ... and the list was short for the first real launch too
    assert len(got) == n - 1
    with ctx.env(RTC_PAIR_JOIN="0"):
        dev_b = api.SketchSet.from_host(sk, ctx.device, k=21)
        want = ctx.mst(dev_b, 0.05)
    assert ctx.diag()["estimates"] == d1["estimates"]
    assert np.array_equal(got, want)
    for e in got[:: max(1, len(got) // 300)]:
        common = int(member[e["preNode"]] @ member[e["sufNode"]])
        assert e["dist"] == api.mst_distance(common, s, s, 21, False)
    # the forest is a spanning tree of minimum weight: no pair is closer than the heaviest edge allows on a cut -- spot check:
    # every genome's best neighbour (most shared hashes) is at least as close as its lightest forest edge
    inter = member[:200] @ member.T
    inter[np.arange(200), np.arange(200)] = 0
    best = inter.max(axis=1)
    lightest = np.full(n, np.inf)
    for e in got:
        lightest[e["preNode"]] = min(lightest[e["preNode"]], e["dist"]); lightest[e["sufNode"]] = min(lightest[e["sufNode"]], e["dist"])
    for g in range(200):
        assert lightest[g] == api.mst_distance(int(best[g]), s, s, 21, False)
    # the pipeline's own loop around rtc_pair_edges_dev follows the same protocol
    from rabbittclust_amd import pipeline
    pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=s, threshold=0.05)
    dev2 = api.SketchSet.from_host(sk, ctx.device, k=21)
    edges, m = pipe.candidate_edges(dev2, 0, n)
    assert m == n * (n - 1) // 2 and edges.shape[0] >= m
    # the context kept the list: the next clustering call starts with it -- no estimate, one launch
    d1 = ctx.diag()  # (the pipeline's list above was a short one of its own: it was handed an estimate too)
    got2 = ctx.mst(dev, 0.05)
    d2 = ctx.diag()
    assert np.array_equal(got2, got) and d2["estimates"] == d1["estimates"] and d2["tiled_tiles"] == d1["tiled_tiles"] + 1
    ctx.close()


def test_pair_tiled_falls_back_when_transposed_copy_exceeds_budget(ctx, oracle):
    """one huge sketch among many small ones inflates the partition-major transposed copy.
    With a tiny budget the tiled path must decline (handled = 0) and the merge kernel must give the
    same counts; algo=2 (tiled only) must then report RTC_ERR_UNSUPPORTED."""
    import os
    from rabbittclust_amd import _lib, api
    rng = np.random.default_rng(17)
    sk = [np.unique(rng.integers(1, 1 << 40, size=60, dtype=np.uint64)) for _ in range(300)]
    sk[7] = np.unique(rng.integers(1, 1 << 40, size=200_000, dtype=np.uint64))
    dev = api.SketchSet.from_host(sk, ctx.device, k=21)
    ref = ctx.pair_common(dev, algo=1).cpu().numpy()
    os.environ["RTC_PAIR_TCOLS_BUDGET"] = "100000"
    _reload_options()
    try:
        got = ctx.pair_common(dev, algo=0).cpu().numpy()
        with pytest.raises(_lib.RtcError) as ei:
            ctx.pair_common(dev, algo=2)
        assert ei.value.status == _lib.RTC_ERR_UNSUPPORTED
    finally:
        del os.environ["RTC_PAIR_TCOLS_BUDGET"]
        _reload_options()
    assert np.array_equal(got, ref)
    assert ref[7, 7] == len(sk[7]) and ref[3, 7] == oracle.common(sk[3], sk[7])


def test_pipeline_step_single_gpu(ctx, oracle):
    from rabbittclust_amd import api, pipeline
    desc = api.synth_family_descs(10, 5, global_seed=21)
    L = 80_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=500, threshold=0.05)
    stats = pipe.step(seq, off)
    host = pipe.last_sketches.to_host()
    flat, start, lens = oracle.to_csr(host)
    want = oracle.mst(flat, start, lens, 21, 0, 0.05)
    got = pipe.last_mst
    assert stats["mst_edges"] == len(want)
    assert np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))


class _LockstepRanks:
    """`world` simulated ranks on one GPU: every rank owns the candidate edges of its triangle row
    range and its own copy of the round state; the per-round key arrays are reduced exactly as the
    RCCL all-reduces would (MIN, MIN, MAX) and every rank applies the device union to the reduced
    arrays -- the states must stay identical."""

    def __init__(self, backends):
        self.b = backends
        self.device = backends[0].device
        self.reduces = 0

    def init(self):
        for b in self.b:
            b.init()

    def _reduce(self, call, out, op):
        import torch
        acc = None
        for b in self.b:
            tmp = torch.empty_like(out)
            call(b, tmp)
            acc = tmp if acc is None else op(acc, tmp)
        out.copy_(acc)
        self.reduces += 1

    def minkey(self, s_fixed, key):
        import torch
        self._reduce(lambda b, o: b.minkey(s_fixed, o), key, torch.minimum)

    def minweight(self, wkey):
        import torch
        self._reduce(lambda b, o: b.minweight(o), wkey, torch.minimum)

    def minedge(self, wkey, ekey):
        import torch
        self._reduce(lambda b, o: b.minedge(wkey, o), ekey, torch.minimum)

    def fetch(self, ekey, ecommon):
        import torch
        self._reduce(lambda b, o: b.fetch(ekey, o), ecommon, torch.maximum)

    def union(self, s_fixed, key, ecommon):
        import torch
        added = [b.union(s_fixed, key, ecommon) for b in self.b]
        assert len(set(added)) == 1
        for b in self.b[1:]:
            assert torch.equal(b.comp, self.b[0].comp)
        return added[0]

    def selected(self):
        return self.b[0].selected()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("fixed", [True, False])
def test_row_sharded_boruvka_equals_single_rank(ctx, oracle, world, fixed):
    """The multi-GPU decomposition on real HIP kernels: triangle row ranges -> per-rank candidate
    edges -> lockstep Boruvka rounds (fixed sizes: ONE reduction per round; else MIN/MIN/MAX) with
    the device union -> same forest weights as the oracle's Kruskal over all pairs, and the SAME
    forest edges as the single-rank run (canonical ids, total order on keys)."""
    import torch
    from rabbittclust_amd import api, pipeline
    desc = api.synth_family_descs(60, 7, global_seed=57)
    L = 60_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    sizes = None if fixed else np.array([300 + 20 * (g % 6) for g in range(len(desc))], dtype=np.uint32)
    sk = ctx.sketch_minhash(seq, off, k=21, size=400, sizes=sizes)
    n = sk.n
    bounds = pipeline.triangle_row_ranges(n, world)
    assert bounds[0] == 0 and bounds[-1] == n
    backends, total_edges = [], 0
    for r in range(world):
        pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=400, threshold=0.05, rank=r, world=world)
        edges, m = pipe.candidate_edges(sk, bounds[r], bounds[r + 1])
        backends.append(pipeline.HipBoruvkaBackend(ctx, sk, edges[:m].clone(), m, False))
        total_edges += m
    pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=400, threshold=0.05)
    s_fixed = pipe.fixed_size(sk)
    assert (s_fixed == 400) == fixed
    ranks = _LockstepRanks(backends)
    sel, rounds = pipeline.boruvka_rounds(ranks, n, None, s_fixed)
    assert ranks.reduces == (rounds if fixed else 3 * rounds)
    got = pipe.finish(sk, sel)
    flat, start, lens = oracle.to_csr(sk.to_host())
    want = oracle.mst(flat, start, lens, 21, 0, 0.05)
    if fixed:
        assert total_edges == len(oracle.candidate_pairs(flat, start, lens))  # equal sizes: the radio filter passes all
    assert len(got) == len(want) and rounds >= 2
    assert np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
    assert _partition(oracle.forest_clusters(got, 0.05, n)) == _partition(oracle.forest_clusters(want, 0.05, n))
    single = ctx.mst(sk, 0.05)
    assert np.array_equal(single, got), "sharded forest differs from the single-rank forest"


class _LoopbackDist:
    """world = 1 stand-in for torch.distributed: the collectives of the multi-GPU step degenerate to
    copies, so pipeline.step runs its N>1 code path (two-part sketch + all-gathers) on one GPU."""

    class _Done:
        def wait(self):
            return True

    class ReduceOp:
        MIN, MAX = "min", "max"

    def all_gather_into_tensor(self, out, inp, async_op=False):
        out.copy_(inp.reshape(-1))
        return self._Done()

    def all_reduce(self, t, op=None):
        return None


def test_pipeline_multi_gpu_code_path_on_one_gpu(ctx, oracle):
    """The distributed branch of MstPipeline.step (sketch in two parts into caller-provided rows, an
    all-gather per part, cost-aware row split) must give the same sketches and forest as the
    single-GPU branch and the oracle."""
    from rabbittclust_amd import api, pipeline
    desc = api.synth_family_descs(12, 5, global_seed=31)
    L = 120_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    ref = pipeline.MstPipeline(ctx, k=21, sketch_size=600, threshold=0.05)
    ref.step(seq, off)
    want_sk = ref.last_sketches.to_host()
    pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=600, threshold=0.05, dist=_LoopbackDist(), rank=0, world=1)
    assert pipe.comm.active
    stats = pipe.step(seq, off)
    got_sk = pipe.last_sketches.to_host()
    assert len(got_sk) == len(want_sk) and all(np.array_equal(a, b) for a, b in zip(got_sk, want_sk))
    flat, start, lens = oracle.to_csr(got_sk)
    want = oracle.mst(flat, start, lens, 21, 0, 0.05)
    assert stats["mst_edges"] == len(want)
    assert np.array_equal(np.sort(pipe.last_mst["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
    # variable sketch sizes through the same path
    sizes = np.array([300 + 10 * (g % 7) for g in range(len(desc))], dtype=np.uint32)
    pipe.step(seq, off, sizes=sizes)
    ref.step(seq, off, sizes=sizes)
    assert all(np.array_equal(a, b) for a, b in zip(pipe.last_sketches.to_host(), ref.last_sketches.to_host()))


@pytest.mark.parametrize("comm", ["native", "torch", "native-strong"])
def test_bench_rccl_path_single_rank(tmp_path, comm):
    """bench.py under torch.distributed.run with one rank and RTC_FORCE_DIST=1: the sketch gather and
    the per-round all-reduces go through RCCL exactly as in the multi-GPU run -- through the C ABI's
    own communicator (--comm native, forced onto RCCL for one rank) or torch.distributed's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTC_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", RTC_COMM_FORCE_RCCL="1")
    strong = comm == "native-strong"  # the strong-scaling form: packed batches through rtc_sketch_minhash_packed_sharded, RCCL forced for the one rank
    comm = comm.split("-")[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", {"native": "29611", "torch": "29612"}[comm] if not strong else "29613", os.path.join(root, "bench.py"), "--gpus", "1",
           "--steps", "1", "--warmup", "1", "--genomes", "200", "--length", "200000", "--no-cpu-baseline", "--comm", comm,
           "--extra-json", str(tmp_path / "x.json")] + (["--scaling", "strong"] if strong else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    line = json.loads(last)
    assert len(last) < 4096
    assert line["n_gpus"] == 1 and line["mst_edges"] > 0 and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["scaling"] == ("strong" if strong else "weak")
    assert line["config"]["collectives"].startswith("rtc_comm (rccl" if comm == "native" else "torch.distributed")
    assert line["config"]["staging"] == "packed" and line["roofline"]["kernel"].startswith("sketch_minhash_packed_kernel")


def test_bench_kssd_mode_small(tmp_path):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "kssd", "--steps", "1", "--warmup", "0",
                        "--genomes", "300", "--length", "300000", "--cpu-sample-genomes", "32"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["dtype"] == "u32" and line["mst_edges"] > 0 and line["roofline"]["kernel"] .startswith("sketch_kssd")
    assert line["cpu_baseline"]["value"] > 0 and "KSSD" in line["cpu_baseline"]["sample"]


def _dense_brute_force(oracle, sk, k, containment, thr, span=100):
    """--dense restated from src/MST.cpp:1333-1352,1468-1530,1703-1713 over the oracle's candidate pairs."""
    flat, start, lens = oracle.to_csr(sk, dtype=sk[0].dtype)
    n = len(sk)
    radio = oracle.lib().orc_mst_radio(thr, k)
    radius = np.array([(1.0 / span) * i for i in range(span)])
    dense = np.zeros((span, n), dtype=np.int64)
    ani = np.zeros(101, dtype=np.uint64)
    for e in oracle.candidate_pairs(flat, start, lens):
        i, j, c = int(e["pre"]), int(e["suf"]), int(e["common"])
        a, b = int(lens[i]), int(lens[j])
        if max(a, b) > radio * min(a, b):
            continue
        d = oracle.lib().orc_mst_distance(c, a, b, k, int(containment))
        t0 = int(np.searchsorted(radius, d, side="left"))
        if t0 < span:
            dense[t0, i] += 1
            dense[t0, j] += 1
        ani[min(int((1.0 - d) * 100.0), 100)] += 1
    return np.cumsum(dense, axis=0).astype(np.int32), ani


@pytest.mark.parametrize("shape", ["fixed", "variable", "kssd"])
def test_mst_dense_histograms_equal_brute_force(ctx, oracle, shape):
    """rtc_mst_dense: density counts and ANI histogram over the candidate pairs, bucketed with the host
    doubles (table per `common` for equal sizes, per-pair otherwise); also through the edge-budget chunk
    path, where the list is contracted between chunks but every pair must still be counted once."""
    import os
    from rabbittclust_amd import api
    rng = np.random.default_rng({"fixed": 41, "variable": 42, "kssd": 43}[shape])
    dt = np.uint32 if shape == "kssd" else np.uint64
    pool = np.unique(rng.integers(1, 1 << 30, size=5000, dtype=np.uint64))
    core = [rng.choice(pool, size=260, replace=False) for _ in range(12)]
    sk = []
    for g in range(300):
        size = 200 if shape == "fixed" else int(rng.integers(60, 260))
        own = core[g % 12][: int(size * rng.uniform(0.3, 1.0))]
        v = np.unique(np.concatenate([own, rng.choice(pool, size=size, replace=False)]))[:size]
        sk.append(np.sort(v).astype(dt))
    k = 22 if shape == "kssd" else 21
    dev = api.SketchSet.from_host(sk, ctx.device, k=k, kind="kssd" if shape == "kssd" else "minhash", width=4 if shape == "kssd" else 8)
    want_dense, want_ani = _dense_brute_force(oracle, sk, k, False, 0.05)
    mst, dense, ani = ctx.mst_dense(dev, 0.05)
    assert np.array_equal(dense, want_dense) and np.array_equal(ani, want_ani)
    assert want_ani.sum() > 5000 and want_dense[-1].max() > 20
    assert np.array_equal(mst, ctx.mst(dev, 0.05))
    os.environ["RTC_EDGE_BUDGET"] = "21000"
    _reload_options()
    try:
        mst2, dense2, ani2 = ctx.mst_dense(dev, 0.05)
    finally:
        del os.environ["RTC_EDGE_BUDGET"]
        _reload_options()
    assert np.array_equal(dense2, want_dense) and np.array_equal(ani2, want_ani) and np.array_equal(mst2, mst)


import os
SOAK_SEEDS = int(os.environ.get("RTC_SOAK_SEEDS", "3"))  # RTC_SOAK_SEEDS=40: a longer walk


@pytest.mark.parametrize("seed", list(range(1, SOAK_SEEDS + 1)))
def test_mst_on_random_sketch_sets(ctx, oracle, seed):
    """Random collections (30 .. 400 sketches of 1 .. 300 hashes out of pools that make ties, empty sketches, exact copies),
    Jaccard and containment, k and threshold per seed: weight multiset bit for bit and the partition at several cuts."""
    from rabbittclust_amd import api
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(30, 400))
    pool = np.unique(rng.integers(1, 1 << 62, size=int(rng.choice([300, 2000, 20000])), dtype=np.uint64))
    smax = int(rng.choice([12, 100, 300]))
    sk = [np.sort(rng.choice(pool, size=min(len(pool), int(rng.integers(1, smax + 1))), replace=False)) for _ in range(n)]
    for _ in range(int(rng.integers(0, 5))):
        sk[int(rng.integers(0, n))] = np.zeros(0, dtype=np.uint64)
    for _ in range(int(rng.integers(0, 5))):
        a, b = rng.integers(0, n, size=2)
        sk[int(a)] = sk[int(b)].copy()
    k = int(rng.choice([17, 21, 25]))
    thr = float(rng.choice([0.01, 0.05, 0.1]))
    dev = api.SketchSet.from_host(sk, ctx.device, k=k)
    flat, start, lens = oracle.to_csr(sk)
    for containment in (False, True):
        got = ctx.mst(dev, thr, is_containment=containment)
        want = oracle.mst(flat, start, lens, k, containment, thr, threads=1)
        assert len(got) == len(want), (seed, containment)
        assert np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64)), (seed, containment)
        for cut in (thr, 0.2, 0.5):
            assert _partition(oracle.forest_clusters(want, cut, n)) == _partition(_clusters_from_edges(got, cut, n)), (seed, containment, cut)
