"""CPU suite: the oracle against public known answers, committed golden fixtures and independent
brute-force restatements.  No GPU."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_murmur3_smhasher_known_answer(oracle):
    # published SMHasher verification value of MurmurHash3_x64_128
    assert oracle.lib().orc_murmur3_smhasher_verification() == 0x6384BA69


def test_murmur3_golden_vectors(oracle):
    g = json.load(open(os.path.join(GOLD, "murmur3_vectors.json")))
    assert g["smhasher_verification"] == "0x6384ba69"
    for v in g["vectors"]:
        assert oracle.murmur3_x64_128(bytes.fromhex(v["key_hex"]), v["seed"]) == (v["h1"], v["h2"])


def test_kmer_hash_is_canonical_and_case_insensitive(oracle):
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rng = np.random.default_rng(1)
    for k in (5, 16, 17, 21, 32):
        kmer = bytes(rng.choice(list(b"ACGT"), size=k).tolist())
        rc = kmer.translate(comp)[::-1]
        h = oracle.kmer_hash(kmer)
        assert h == oracle.kmer_hash(rc) == oracle.kmer_hash(kmer.lower())
        canon = min(kmer, rc)
        h1, _ = oracle.murmur3_x64_128(canon, 42)
        assert h == (h1 if k > 16 else h1 & 0xFFFFFFFF)


def test_minhash_sketch_is_bottom_s_of_all_kmer_hashes(oracle):
    rng = np.random.default_rng(2)
    g = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=4000, p=[.245, .245, .245, .245, .02])
    off = np.array([0, len(g)], dtype=np.uint64)
    k, s = 11, 64
    got = oracle.sketch_minhash_batch(g, off, k, s)[0]
    raw = g.tobytes()
    hs = {oracle.kmer_hash(raw[i:i + k]) for i in range(len(raw) - k + 1) if b"N" not in raw[i:i + k]}
    assert got.tolist() == sorted(hs)[:s]


def test_sketch_fixture_frozen(oracle):
    fx = np.load(os.path.join(GOLD, "sketch_fixture.npz"), allow_pickle=True)
    L = int(fx["L"])
    genomes = [oracle.synth_genome(int(f), int(m), int(t), L, int(ne)) for f, m, t, ne in fx["descs"]]
    seq = np.concatenate(genomes)
    off = np.arange(len(genomes) + 1, dtype=np.uint64) * L
    mh = oracle.sketch_minhash_batch(seq, off, 21, 400, threads=1)
    for a, b in zip(mh, fx["minhash"]):
        assert np.array_equal(a, b)
    for g, b in zip(genomes, fx["kssd"]):
        assert np.array_equal(oracle.kssd_sketch(g, 21, 3), b)
    flat, start, lens = oracle.to_csr(mh)
    mst = oracle.mst(flat, start, lens, 21, 0, 0.05, threads=1)
    assert np.array_equal(mst["dist"].view(np.uint64), fx["mst"]["dist"].view(np.uint64))
    assert np.array_equal(mst["pre"], fx["mst"]["pre"]) and np.array_equal(mst["suf"], fx["mst"]["suf"])


def test_kssd_shuffle_table_matches_fixture(oracle):
    fx = np.load(os.path.join(GOLD, "kssd_shuffle_hs6.npz"))
    sd = oracle.kssd_shuffle_dim(6)
    assert np.array_equal(sd[:64], fx["head"])
    kept = np.nonzero(sd < 4096)[0]
    assert np.array_equal(kept.astype(np.uint32), fx["dim_id"])
    assert np.array_equal(sd[kept].astype(np.uint16), fx["rank"])
    assert len(np.unique(sd)) == 1 << 24


def test_kssd_sketch_matches_bruteforce_restatement(oracle):
    """Independent pure-Python restatement of src/SketchInfo.cpp:1126-1165 on a short genome."""
    rng = np.random.default_rng(3)
    g = rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=30000)
    k, dr = 21, 3
    p = oracle.kssd_params(k, dr)
    sd = oracle.kssd_shuffle_dim(p.half_subk)
    half_k, hs = p.half_k, p.half_subk
    K = 2 * half_k
    tupmask = (1 << (4 * half_k)) - 1
    hoc = half_k - hs
    domask = (tupmask >> (4 * hoc)) << (2 * hoc)
    undomask = (tupmask ^ domask) & tupmask
    und1 = undomask & (tupmask >> ((half_k + hs) * 2))
    und0 = undomask ^ und1
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    tup = rvs = 0
    base = 1
    out = set()
    for ch in g.tolist():
        b = code.get(ch, -1)
        if b < 0:
            base, tup, rvs = 1, 0, 0
            continue
        tup = ((tup << 2) | b) & tupmask
        rvs = (rvs >> 2) + ((b ^ 3) << (4 * half_k - 2))
        base += 1
        if base > K:
            uni = min(tup, rvs)
            dim = (uni & domask) >> (2 * hoc)
            r = int(sd[dim])
            if r >= p.dim_end:
                continue
            out.add(((((uni & und0) | ((uni & und1) << (2 * K - 4 * hoc))) >> (4 * dr)) | r) & 0xFFFFFFFF)
    assert oracle.kssd_sketch(g, k, dr).tolist() == sorted(out)


def test_distances_and_radio(oracle):
    L = oracle.lib()
    assert L.orc_mst_radio(0.05, 21) == 4       # (int)(2*exp(0.05*20) - 1) = 4.436 -> 4
    assert L.orc_mst_distance(1000, 1000, 1000, 21, 0) == 0.0
    assert L.orc_mst_distance(0, 1000, 1000, 21, 0) == 1.0
    j = 500 / 1500
    assert L.orc_mst_distance(500, 1000, 1000, 21, 0) == -(1.0 / 21) * math.log((2.0 * j) / (1.0 + j))
    assert L.orc_mst_distance(300, 600, 1000, 21, 1) == -(1.0 / 21) * math.log(300 / 600)
    # greedy clamps at 1, MST does not
    assert L.orc_greedy_distance(1, 100000, 100000, 5, 0) == 1.0
    assert L.orc_mst_distance(1, 100000, 100000, 5, 0) > 1.0
    assert L.orc_kssd_greedy_distance(0, 10, 10, 21) == 1.0


def test_tune_parameters_table(oracle):
    # 1 Mbp FASTA (1,012,520 B): user -k 21 is replaced by 17; 5 Mbp keeps 21
    r = oracle.tune_parameters(0, 1, 0, 1, 21, 0.05, 1000, 1000, 1012520, 1012520, 1012520)
    assert (r.kmer_size, r.ok) == (17, 1)
    r = oracle.tune_parameters(0, 1, 0, 1, 21, 0.05, 1000, 1000, 5062520, 5062520, 5062520)
    assert (r.kmer_size, r.ok) == (21, 1)
    r = oracle.tune_parameters(0, 0, 0, 0, 19, 0.05, 1000, 1000, 5062520, 5062520, 5062520)
    assert r.kmer_size == 18
    # greedy without -s/-c defaults to containment with compress = avg/1000
    r = oracle.tune_parameters(1, 1, 0, 0, 21, 0.05, 1000, 1000, 5062520, 5062520, 5062520)
    assert (r.is_containment, r.contain_compress) == (1, 5062)
    # -s and -c together are rejected; too large a threshold is rejected
    assert oracle.tune_parameters(0, 1, 1, 1, 21, 0.05, 1000, 1000, 5062520, 5062520, 5062520).ok == 0
    assert oracle.tune_parameters(0, 1, 0, 1, 21, 0.9, 1000, 1000, 5062520, 5062520, 5062520).ok == 0


def _brute_edges(sk, k, containment, thr, oracle):
    n = len(sk)
    radio = oracle.lib().orc_mst_radio(thr, k)
    out = []
    for i in range(n):
        for j in range(i):
            c = len(np.intersect1d(sk[i], sk[j]))
            if c == 0 or len(sk[i]) == 0 or len(sk[j]) == 0:
                continue
            if max(len(sk[i]), len(sk[j])) > radio * min(len(sk[i]), len(sk[j])):
                continue
            out.append((oracle.lib().orc_mst_distance(c, len(sk[i]), len(sk[j]), k, int(containment)), i, j))
    return out


@pytest.mark.parametrize("containment", [False, True])
@pytest.mark.parametrize("threads", [1, 3])
def test_oracle_mst_against_bruteforce_kruskal(oracle, containment, threads):
    rng = np.random.default_rng(4)
    pool = np.unique(rng.integers(1, 1 << 60, size=900, dtype=np.uint64))
    sk = [np.sort(rng.choice(pool, size=int(rng.integers(0, 120)), replace=False)) for _ in range(61)]
    flat, start, lens = oracle.to_csr(sk)
    got = oracle.mst(flat, start, lens, 21, containment, 0.05, threads=threads)
    edges = sorted(_brute_edges(sk, 21, containment, 0.05, oracle))
    parent = list(range(len(sk)))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    want = []
    for d, i, j in edges:
        a, b = find(i), find(j)
        if a != b:
            parent[a] = b
            want.append(d)
    assert sorted(got["dist"].tolist()) == sorted(want)
    # candidate pairs = every pair with a non-empty intersection
    cand = oracle.candidate_pairs(flat, start, lens)
    assert len(cand) == sum(1 for i in range(len(sk)) for j in range(i) if len(np.intersect1d(sk[i], sk[j])))


def test_unionfind_and_kruskal_against_reference_fixture(oracle):
    fx = np.load(os.path.join(GOLD, "unionfind.npz"))
    edges = np.zeros(len(fx["xs"]), dtype=oracle.EDGE_DT)
    edges["pre"], edges["suf"], edges["dist"] = fx["xs"], fx["ys"], np.arange(len(fx["xs"]))
    keep = edges[edges["pre"] != edges["suf"]]
    out = np.zeros(len(keep) + 1, dtype=oracle.EDGE_DT)
    m = oracle.lib().orc_kruskal(keep.ctypes.data_as(C.c_void_p), C.c_uint64(len(keep)), C.c_int(200),
                                 out.ctypes.data_as(C.c_void_p))
    # the forest connects exactly what the reference UnionFind connected
    parent = list(range(200))

    def find(x):
        while parent[x] != x:
            x = parent[x]
        return x
    for e in out[:m]:
        parent[find(int(e["pre"]))] = find(int(e["suf"]))
    roots = fx["roots"]
    for a in range(200):
        for b in (0, 57, 199):
            assert (find(a) == find(b)) == (roots[a] == roots[b])
    assert 200 - m == int(fx["size"])


def test_greedy_oracle_basic_invariants(oracle):
    rng = np.random.default_rng(5)
    anc = [np.unique(rng.integers(0, 1 << 50, size=300, dtype=np.uint64)) for _ in range(8)]
    sk = []
    for a in anc:
        for m in range(4):
            v = a.copy()
            idx = rng.choice(len(v), size=10 * m, replace=False)
            v[idx] = rng.integers(0, 1 << 50, size=len(idx), dtype=np.uint64)
            sk.append(np.unique(v))
    flat, start, lens = oracle.to_csr(sk)
    n, rep = oracle.greedy_minhash(flat, start, lens, 300, 21, False, 0.05)
    assert n == 8
    for i, r in enumerate(rep):
        assert rep[r] == r and r <= i and r // 4 == i // 4


# ---- an independent pure-Python twin of the greedy restatement (dict-based inverted index, the
# reference's own loop order) -- every other restated function has one; src/greedy.cpp:986-1399, :566-899
def _py_greedy(sk, size_cfg, k, containment, thr, kssd=False):
    import math
    n = len(sk)
    rep_of = list(range(n))
    index = {}                                    # hash -> [rep ids in insertion order]
    for h in sk[0].tolist():
        index.setdefault(h, []).append(0)
    x = math.exp(-thr * k)
    jmin = x / (2.0 - x)
    fast = False
    fixed_min = 0
    if not kssd:
        fixed = int(size_cfg[0])
        fast = (not containment) and all(int(size_cfg[i]) == fixed for i in range(1, min(100, n)))
        if fast:
            fixed_min = int(math.ceil(jmin * (2 * fixed) / (1.0 + jmin)))
    for j in range(1, n):
        q = sk[j].tolist()
        size_ref = len(q)
        cnt, touched = {}, []
        for h in q:                               # first-touch order = the -t 1 visiting order
            for r in index.get(h, ()):
                if r not in cnt:
                    cnt[r] = 1
                    touched.append(r)
                else:
                    cnt[r] += 1
        best_rep, best_common, best_dist, best_jac = -1, -1, float("inf"), -1.0
        for r in touched:
            common = cnt[r]
            if kssd:
                size_q = len(sk[r])
                if common < int(math.ceil(jmin * (size_ref + size_q) / (1.0 + jmin))):
                    continue
                den = size_ref + size_q - common
                jac = 1.0 if den == 0 else common / den
                if jac > best_jac:
                    best_jac, best_rep = jac, r
                continue
            size_q = int(size_cfg[r])
            if fast:
                cmin = fixed_min
            elif containment:
                cmin = int(math.ceil(jmin * min(size_ref, size_q)))
            else:
                cmin = int(math.ceil(jmin * (size_ref + size_q) / (1.0 + jmin)))
            if common < cmin:
                continue
            if fast:
                if common > best_common:
                    best_common, best_rep = common, r
            else:
                den = min(size_ref, size_q) if containment else size_ref + size_q - common
                if den == 0:
                    dist = 1.0 if containment else 0.0
                else:
                    jac = common / den
                    if jac >= 1.0:
                        dist = 0.0
                    elif jac <= 0.0:
                        dist = 1.0
                    else:
                        dist = min(-math.log(2.0 * jac / (1.0 + jac)) / k, 1.0)
                if dist <= thr and dist < best_dist:
                    best_dist, best_rep = dist, r
        if best_rep >= 0:
            rep_of[j] = best_rep
        else:
            for h in q:
                index.setdefault(h, []).append(j)
    return sum(1 for i in range(n) if rep_of[i] == i), rep_of


def _greedy_families(rng, n_fam, per, size, ragged, dtype=np.uint64, bits=50):
    sk = []
    for _ in range(n_fam):
        anc = np.unique(rng.integers(0, 1 << bits, size=2 * size, dtype=np.uint64))[:size]
        for m in range(per):
            s = size if not ragged else int(size * rng.uniform(0.3, 1.0))
            v = anc[:s].copy()
            if m % 3:
                idx = rng.choice(len(v), size=int(len(v) * 0.5 * rng.uniform(0, 1)), replace=False)
                v[idx] = rng.integers(0, 1 << bits, size=len(idx), dtype=np.uint64)
            sk.append(np.unique(v).astype(dtype))
    order = rng.permutation(len(sk))
    return [sk[i] for i in order]


@pytest.mark.parametrize("mode", ["fixed", "containment", "variable", "kssd"])
def test_greedy_oracle_equals_python_twin(oracle, mode):
    rng = np.random.default_rng({"fixed": 71, "containment": 72, "variable": 73, "kssd": 74}[mode])
    if mode == "fixed":
        sk = _greedy_families(rng, 30, 6, 200, False)
        cfg = np.full(len(sk), 200, dtype=np.uint32)
        flat, start, lens = oracle.to_csr(sk)
        for thr in (0.02, 0.05):
            want_n, want = _py_greedy(sk, cfg, 21, False, thr)
            got_n, got = oracle.greedy_minhash(flat, start, lens, cfg, 21, False, thr)
            assert got_n == want_n and got.tolist() == want
        assert 1 < got_n < len(sk)
    elif mode in ("containment", "variable"):
        sk = _greedy_families(rng, 25, 6, 300, True)
        cfg = np.array([max(len(s), 100) for s in sk], dtype=np.uint32)
        flat, start, lens = oracle.to_csr(sk)
        cont = mode == "containment"
        want_n, want = _py_greedy(sk, cfg, 21, cont, 0.05)
        got_n, got = oracle.greedy_minhash(flat, start, lens, cfg, 21, cont, 0.05)
        assert got_n == want_n and got.tolist() == want and 1 < got_n < len(sk)
    else:
        sk = _greedy_families(rng, 25, 6, 250, True, dtype=np.uint32, bits=30)
        sk.sort(key=lambda a: -len(a))
        flat, start, lens = oracle.to_csr(sk, dtype=np.uint32)
        want_n, want = _py_greedy(sk, None, 22, False, 0.05, kssd=True)
        got_n, got = oracle.greedy_kssd(flat, start, lens, 22, 0.05)
        assert got_n == want_n and got.tolist() == want and 1 < got_n < len(sk)


def test_sketch_threshold_test_word_bounds():
    """The sketch kernel's one-compare threshold test (csrc/rtc_sketch_minhash.hip, hash_test_word): with a, b the two
    fmix64 halves before their last multiply by C, w = hi32((a + b) * C mod 2^64) + 1 must equal hi32(hash) + {0, 1, 2}
    (mod 2^32) for hash = fin(a * C) + fin(b * C), fin(f) = f ^ f >> 33 -- random halves and halves built so that
    hi(f1) + hi(f2) wraps or sits at 2^32 - 1 / 2^32 - 2 (the cases the slack of 2 exists for)."""
    M64, M32 = (1 << 64) - 1, (1 << 32) - 1
    Cm = 0xc4ceb9fe1a85ec53
    Cinv = pow(Cm, -1, 1 << 64)
    rng = np.random.default_rng(7)

    def check(a, b):
        f1, f2 = a * Cm & M64, b * Cm & M64
        h = ((f1 ^ f1 >> 33) + (f2 ^ f2 >> 33)) & M64
        w = ((((a + b) & M64) * Cm & M64) >> 32) + 1 & M32
        d = (w - (h >> 32)) & M32
        assert d in (0, 1, 2), (hex(a), hex(b), d)
        return d

    seen = set()
    for a, b in rng.integers(0, 1 << 64, size=(20000, 2), dtype=np.uint64).tolist():
        seen.add(check(a, b))
    # products with chosen high words: f1 = x, f2 = y  =>  a = x * C^-1, b = y * C^-1
    for hi_sum in (M32, M32 - 1, 0, 1, (1 << 32), (1 << 32) + 1):
        for _ in range(2000):
            h1 = int(rng.integers(0, 1 << 32))
            h2 = (hi_sum - h1) & M32
            for lo1, lo2 in ((int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))), (M32, M32), (0, 0), (M32, 1)):
                seen.add(check((h1 << 32 | lo1) * Cinv & M64, (h2 << 32 | lo2) * Cinv & M64))
    assert seen == {0, 1, 2}
