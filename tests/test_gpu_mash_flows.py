"""GPU: the dense estimator loops of the reference -- modifyMST (src/MST.cpp:809-1018) and the legacy greedyCluster
(src/greedy.cpp:285-351) -- behind rtc_mst_mash / rtc_greedy_mash and the command lines that reach them
(--inverted-index=false, clust-greedy --append on MinHash sketches), plus the MinHash cluster_state.bin
(MinHashClusterState::save / ::load, src/greedy.cpp:2134-2302).  Checked against Python restatements written from
the reference and the published Mash estimator: parity-UNPINNED like the k-mer hash, because
RabbitSketch's MinHash::distance() is absent from the reference tree."""
import math
import os
import struct

import numpy as np
import pytest

from test_gpu_cli import BIN, _parse_clusters, _partition, _run, _write_family_fastas
from test_gpu_cli import _read_hash_sketch as _read_hash_sketch_hdr


def _read_hash_sketch(folder):
    return _read_hash_sketch_hdr(folder)[1]

pytestmark = pytest.mark.gpu


# ---- restatements ------------------------------------------------------------------------------------------
def _mash_counts(a, b, s):
    """Mash's merge: stop after s elements of the union (or when both lists are used up)."""
    i = j = c = d = 0
    while d < s and i < len(a) and j < len(b):
        if a[i] < b[j]:
            i += 1
        elif b[j] < a[i]:
            j += 1
        else:
            c += 1; i += 1; j += 1
        d += 1
    if d < s:
        d += min((len(a) - i) + (len(b) - j), s - d)
    return c, d


def _mash_distance(a, b, s, k):
    c, d = _mash_counts(a, b, s)
    j = c / d if d else 0.0
    if j == 0.0:
        return 1.0
    if j == 1.0:
        return 0.0
    return min(1.0, -math.log(2.0 * j / (1.0 + j)) / k)


def _contain_distance(a, b, k):
    c = len(np.intersect1d(a, b, assume_unique=True))
    mn = min(len(a), len(b))
    cj = c / mn if mn else 0.0
    return 1.0 if cj == 0.0 else (0.0 if cj == 1.0 else -(1.0 / k) * math.log(cj))  # the in-tree form (src/MST.cpp:1295,1515)


def _dist_matrix(sk, s, k, containment):
    n = len(sk)
    D = np.ones((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            D[i, j] = D[j, i] = _contain_distance(sk[i], sk[j], k) if containment else _mash_distance(sk[i], sk[j], s, k)
    return D


def _kruskal_weights(D, start_index=0):
    n = len(D)
    edges = sorted((D[i, j], i, j) for i in range(n) for j in range(max(i + 1, start_index), n))
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    w = []
    for d, i, j in edges:
        a, b = find(i), find(j)
        if a != b:
            parent[a] = b
            w.append(d)
    return np.array(w)


def _components(n, edges, thr):
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for i, j, d in edges:
        if d <= thr:
            parent[find(i)] = find(j)
    comp = {}
    for v in range(n):
        comp.setdefault(find(v), []).append(v)
    return sorted(tuple(c) for c in comp.values())


def _greedy_legacy(sk, s, k, thr, containment):
    """greedyCluster: nearest representative within thr (earliest of equals) or a new cluster."""
    reps, rep_of = [0], [0]
    for q in range(1, len(sk)):
        best, br = None, -1
        for r in reps:
            d = _contain_distance(sk[r], sk[q], k) if containment else _mash_distance(sk[r], sk[q], s, k)
            if d <= thr and (best is None or d < best):
                best, br = d, r
        if br >= 0:
            rep_of.append(br)
        else:
            rep_of.append(q); reps.append(q)
    return rep_of


def _clusters_of(rep_of):
    cl, cid = [], {}
    for i, r in enumerate(rep_of):
        if r == i:
            cid[i] = len(cl); cl.append([i])
    for i, r in enumerate(rep_of):
        if r != i:
            cl[cid[r]].append(i)
    return cl


def _sketches(ctx, oracle, n_fam, per, L, s, seed, sizes=None):
    from rabbittclust_amd import api
    desc = api.synth_family_descs(n_fam, per, global_seed=seed, max_rate=0.06)
    n = len(desc)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sk = ctx.sketch_minhash(seq, off, k=21, size=s, sizes=sizes)
    ctx.sync()
    return sk, sk.to_host()


# ---- C ABI ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("containment", [False, True])
def test_mst_mash_equals_dense_loop_restatement(ctx, oracle, containment, monkeypatch):
    s, k = 300, 21
    sizes = np.array([200 + 20 * (g % 7) for g in range(36)], dtype=np.uint32) if containment else None
    sk, host = _sketches(ctx, oracle, 6, 6, 120_000, s, 71, sizes)
    n = len(host)
    D = _dist_matrix(host, s, k, containment)
    want = _kruskal_weights(D)
    mst = ctx.mst_mash(sk, s, is_containment=containment)
    assert len(mst) == n - 1 and np.all(mst["preNode"] < mst["sufNode"])          # a spanning TREE, EdgeInfo{i < j}
    assert np.array_equal(np.sort(mst["dist"]).view(np.uint64), np.sort(want).view(np.uint64))
    for e in mst:
        assert e["dist"] == D[e["preNode"], e["sufNode"]]
    edges = [(int(e["preNode"]), int(e["sufNode"]), float(e["dist"])) for e in mst]
    full = [(i, j, D[i, j]) for i in range(n) for j in range(i + 1, n)]
    for thr in (0.02, 0.05, 0.5):
        assert _components(n, edges, thr) == _components(n, full, thr)
    # the --append form: only pairs with j >= start_index
    part = ctx.mst_mash(sk, s, is_containment=containment, start_index=20)
    assert np.array_equal(np.sort(part["dist"]).view(np.uint64), np.sort(_kruskal_weights(D, 20)).view(np.uint64))
    assert np.all(part["sufNode"] >= 20)
    # the same forest when the edge list has to be contracted between row chunks
    # (the 630 pairs of these 36 sketches fit the smallest budget the library accepts: a set of 72, 2 556 pairs, does not)
    sizes2 = np.array([200 + 20 * (g % 7) for g in range(72)], dtype=np.uint32) if containment else None
    sk2, host2 = _sketches(ctx, oracle, 12, 6, 120_000, s, 73, sizes2)
    plain = ctx.mst_mash(sk2, s, is_containment=containment)
    with ctx.env(RTC_EDGE_BUDGET="1024"):  # (the library reads its switches at context creation: ctx.env makes this context read them again)
        c0 = ctx.diag()["contractions"]
        again = ctx.mst_mash(sk2, s, is_containment=containment)
        assert ctx.diag()["contractions"] > c0, "the contraction path did not run"
    assert np.array_equal(again, plain) and len(plain) == 71
    assert np.array_equal(np.sort(plain["dist"]).view(np.uint64), np.sort(_kruskal_weights(_dist_matrix(host2, s, k, containment))).view(np.uint64))
    # --dense by-products: EVERY pair counts (src/MST.cpp:868-879)
    mst2, dense, ani = ctx.mst_mash(sk, s, is_containment=containment, span=100)
    assert np.array_equal(mst2, mst)
    radius = [i / 100.0 * 1.0 for i in range(100)]
    radius = [(1.0 / 100) * i for i in range(100)]
    want_d = np.zeros((100, n), dtype=np.int64)
    want_a = np.zeros(101, dtype=np.uint64)
    for i in range(n):
        for j in range(i + 1, n):
            t0 = int(np.searchsorted(radius, D[i, j], side="left"))
            if t0 < 100:
                want_d[t0, i] += 1; want_d[t0, j] += 1
            want_a[min(100, max(0, int((1.0 - D[i, j]) * 100.0)))] += 1
    assert np.array_equal(dense, np.cumsum(want_d, axis=0)) and np.array_equal(ani, want_a)


@pytest.mark.parametrize("containment", [False, True])
def test_greedy_mash_equals_legacy_loop_restatement(ctx, oracle, containment):
    s, k = 300, 21
    sizes = np.array([220 + 15 * (g % 5) for g in range(40)], dtype=np.uint32) if containment else None
    sk, host = _sketches(ctx, oracle, 8, 5, 120_000, s, 72, sizes)
    for thr in (0.03, 0.06):
        ncl, rep = ctx.greedy_mash(sk, thr, s, is_containment=containment)
        want = _greedy_legacy(host, s, k, thr, containment)
        assert rep.tolist() == want and ncl == sum(1 for i, r in enumerate(want) if r == i)
    assert 8 <= ncl < 40


# ---- command lines -------------------------------------------------------------------------------------------
def _folder(d):
    return [os.path.join(d, x) for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))][0]


def test_cli_index_off_runs_the_dense_loops(oracle, tmp_path):
    """clust-mst / clust-greedy --inverted-index=false: modifyMST and greedyCluster on the same hash.sketch"""
    tmp = str(tmp_path)
    L, s = 2_000_000, 400
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 4, 4, L, seed=81)
    dm = os.path.join(tmp, "m"); os.makedirs(dm)
    out = os.path.join(dm, "m.out")
    _run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", str(s), "-d", "0.05", "-t", "4", "--inverted-index=false", "-o", out], dm)
    folder = _folder(dm)
    sk = _read_hash_sketch(folder)
    n = len(sk)
    D = _dist_matrix(sk, s, 21, False)
    raw = open(os.path.join(folder, "edge.mst"), "rb").read()
    (ne,) = struct.unpack_from("<Q", raw, 0)
    edges = [struct.unpack_from("<iid", raw, 8 + 16 * e) for e in range(ne)]
    assert ne == n - 1 and all(i < j and d == D[i, j] for i, j, d in edges)
    assert np.array_equal(np.sort([d for _, _, d in edges]), np.sort(_kruskal_weights(D)))
    full = [(i, j, D[i, j]) for i in range(n) for j in range(i + 1, n)]
    assert _partition(_parse_clusters(out)) == _partition(_components(n, full, 0.05))
    # the index path on the same genomes gives the same partition here (its set-Jaccard distances differ in the last digits only)
    out2 = os.path.join(dm, "i.out")
    _run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", str(s), "-d", "0.05", "-t", "4", "-e", "-o", out2], dm)
    assert _partition(_parse_clusters(out2)) == _partition(_parse_clusters(out))
    # clust-greedy: greedyCluster in list order
    dg = os.path.join(tmp, "g"); os.makedirs(dg)
    outg = os.path.join(dg, "g.out")
    _run([os.path.join(BIN, "clust-greedy"), "-l", "-i", lst, "-k", "21", "-s", str(s), "-d", "0.05", "-t", "4", "--inverted-index=false", "-e", "-o", outg], dg)
    assert _parse_clusters(outg) == _clusters_of(_greedy_legacy(sk, s, 21, 0.05, False))


def _read_mh_state(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"MINHASH\0"
    thr, k, s, cont = struct.unpack_from("<dii?", raw, 8)
    pos = 8 + 17
    (nr,) = struct.unpack_from("<Q", raw, pos); pos += 8
    reps = list(struct.unpack_from(f"<{nr}i", raw, pos)); pos += 4 * nr
    (ns,) = struct.unpack_from("<Q", raw, pos); pos += 8
    sketches = []
    for _ in range(ns):
        gid, length, hc = struct.unpack_from("<iQQ", raw, pos); pos += 20
        h = np.frombuffer(raw, dtype="<u8", count=hc, offset=pos).copy(); pos += 8 * hc
        (nl,) = struct.unpack_from("<Q", raw, pos); pos += 8
        name = raw[pos:pos + nl].decode(); pos += nl
        sketches.append((gid, length, h, name))
    (nc,) = struct.unpack_from("<Q", raw, pos); pos += 8
    clusters = []
    for _ in range(nc):
        (m,) = struct.unpack_from("<Q", raw, pos); pos += 8
        clusters.append(list(struct.unpack_from(f"<{m}i", raw, pos))); pos += 4 * m
    (ni,) = struct.unpack_from("<Q", raw, pos); pos += 8
    index = {}
    for _ in range(ni):
        h, m = struct.unpack_from("<QQ", raw, pos); pos += 16
        index[h] = list(struct.unpack_from(f"<{m}i", raw, pos)); pos += 4 * m
    assert pos == len(raw)
    return dict(threshold=thr, k=k, s=s, containment=cont, reps=reps, sketches=sketches, clusters=clusters, index=index)


def test_clust_greedy_minhash_append_and_cluster_state(oracle, tmp_path):
    tmp = str(tmp_path)
    L, s, thr = 2_000_000, 400, 0.05
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 4, 4, L, seed=82)
    perm = [0, 5, 10, 1, 4, 8, 9, 2, 3, 6, 7, 13, 11, 12, 14, 15]   # family 3 arrives with the appended genomes only
    first, second = perm[:7], perm[7:]
    la, lb = os.path.join(tmp, "a.txt"), os.path.join(tmp, "b.txt")
    open(la, "w").write("\n".join(paths[i] for i in first) + "\n")
    open(lb, "w").write("\n".join(paths[i] for i in second) + "\n")
    # ---- batch run with --save-rep: MinHashInitialClusterWithState + MinHashClusterState::save ----
    da = os.path.join(tmp, "a"); os.makedirs(da)
    outa = os.path.join(da, "a.out")
    err = _run([os.path.join(BIN, "clust-greedy"), "-l", "-i", la, "-k", "21", "-s", str(s), "-d", str(thr), "-t", "4", "--save-rep", "-o", outa], da)
    folder = _folder(da)
    state = os.path.join(folder, "cluster_state.bin")
    assert "saved cluster state (with inverted index) to:" in err and os.path.exists(state)
    st = _read_mh_state(state)
    sk_a = _read_hash_sketch(folder)
    cl_a = _parse_clusters(outa)
    assert (st["threshold"], st["k"], st["s"], st["containment"]) == (thr, 21, s, False)
    assert st["clusters"] == cl_a and st["reps"] == [c[0] for c in cl_a]
    assert [x[3] for x in st["sketches"]] == [paths[i] for i in first] and all(np.array_equal(x[2], sk_a[i]) for i, x in enumerate(st["sketches"]))
    want_index = {}
    for r, g in enumerate(st["reps"]):
        for h in sk_a[g].tolist():
            want_index.setdefault(h, []).append(r)
    assert st["index"] == want_index
    # ---- --append with the stored state: MinHashIncrementalCluster over the folder's sketches ----
    outb = os.path.join(tmp, "ab.out")
    err = _run([os.path.join(BIN, "clust-greedy"), "-l", "--presketched", folder, "--append", lb, "-d", str(thr), "-t", "4", "--save-rep", "-o", outb], tmp)
    assert "Incremental Update Mode (MinHash)" in err and "Successfully loaded %d representatives" % len(cl_a) in err
    new = oracle.sketch_minhash_batch(np.concatenate([seqs[i] for i in second]), np.arange(len(second) + 1, dtype=np.uint64) * np.uint64(L), 21, s)
    sets = [set(x.tolist()) for x in sk_a] + [set(x.tolist()) for x in new]
    clusters = [list(c) for c in cl_a]
    rep_sets = [sets[c[0]] for c in cl_a]
    x = math.exp(-thr * 21); jmin = x / (2.0 - x)
    for q in range(len(first), len(sets)):
        best, br = None, -1
        for r, rs in enumerate(rep_sets):
            cm = len(sets[q] & rs)
            if cm == 0 or cm < int(jmin * (len(sets[q]) + len(rs)) / (1.0 + jmin)):
                continue
            den = len(sets[q]) + len(rs) - cm
            jac = cm / den
            d = 0.0 if jac >= 1.0 else min(1.0, -math.log(2.0 * jac / (1.0 + jac)) / 21)
            if d <= thr and (best is None or d < best):
                best, br = d, r
        if br >= 0:
            clusters[br].append(q)
        else:
            clusters.append([]); rep_sets.append(sets[q])
    assert _parse_clusters(outb) == clusters and len(clusters) > len(cl_a)
    st2 = _read_mh_state(state)   # the state was written back with the new genomes
    assert st2["clusters"] == clusters and len(st2["sketches"]) == 16 and len(st2["reps"]) == len(clusters)
    # ---- --append WITHOUT a state: everything together, size-sorted, legacy greedyCluster ----
    db = os.path.join(tmp, "b"); os.makedirs(db)
    _run([os.path.join(BIN, "clust-greedy"), "-l", "-i", la, "-k", "21", "-s", str(s), "-d", str(thr), "-t", "4", "-o", os.path.join(db, "a.out")], db)
    folder2 = _folder(db)
    dc = os.path.join(tmp, "c"); os.makedirs(dc)
    outc = os.path.join(dc, "c.out")
    err = _run([os.path.join(BIN, "clust-greedy"), "-l", "--presketched", folder2, "--append", lb, "-d", str(thr), "-t", "4", "-o", outc], dc)
    assert "---use the Mash distance (fixed-sketch-size), the sketch size is: %d" % s in err
    allsk = list(_read_hash_sketch(folder2)) + list(new)          # equal genome lengths: the order is (id ascending) stable
    ids = list(range(len(first))) + list(range(len(second)))
    order = sorted(range(16), key=lambda q: ids[q])               # cmpGenomeSize on equal lengths: by id, pre before appended on ties
    sorted_sk = [allsk[q] for q in order]
    assert _parse_clusters(outc) == _clusters_of(_greedy_legacy(sorted_sk, s, 21, thr, False))
    saved = _read_hash_sketch(_folder(dc))
    assert len(saved) == 16 and all(np.array_equal(a, b) for a, b in zip(saved, sorted_sk))
