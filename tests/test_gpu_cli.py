"""GPU end-to-end: the clust-mst / clust-greedy command lines on FASTA files, checked against the
oracle run on the same bytes (sketch file contents, MST weights, cluster partition, resume flows)."""
import math
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "rabbittclust_amd", "bin")


def _write_family_fastas(oracle, tmp, n_fam, per, L, seed=1, two_records=False):
    from rabbittclust_amd import api
    desc = api.synth_family_descs(n_fam, per, global_seed=seed)
    paths, seqs = [], []
    for g, d in enumerate(desc):
        s = oracle.synth_genome(int(d["fam_seed"]), int(d["mut_seed"]), int(d["mut_thr"]), L)
        p = os.path.join(tmp, f"g{g:03d}.fna")
        raw = s.tobytes()
        with open(p, "wb") as f:
            if two_records and g % 2:
                cut = L // 3
                recs = [(f">g{g}_a synthetic family {g // per}\n", raw[:cut]), (f">g{g}_b\n", raw[cut:])]
            else:
                recs = [(f">g{g} synthetic family {g // per}\n", raw)]
            for hdr, body in recs:
                f.write(hdr.encode())
                for i in range(0, len(body), 80):
                    f.write(body[i:i + 80] + b"\n")
        paths.append(p)
        seqs.append(s)
    lst = os.path.join(tmp, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    return lst, paths, seqs


def _parse_clusters(path):
    clusters, cur = [], None
    for ln in open(path):
        if ln.startswith("the cluster"):
            cur = []
            clusters.append(cur)
        elif ln.startswith("\t"):
            cur.append(int(ln.split("\t")[2]))
    return clusters


def _partition(cl):
    return sorted(tuple(sorted(c)) for c in cl)


def _run(args, cwd, env=None):
    r = subprocess.run(args, cwd=cwd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stderr


def _read_hash_sketch(folder):
    raw = open(os.path.join(folder, "hash.sketch"), "rb").read()
    fid, k, cont, val = struct.unpack_from("<ii?i", raw, 0)
    pos, out = 13, []
    while pos < len(raw):
        (m,) = struct.unpack_from("<Q", raw, pos); pos += 8
        out.append(np.frombuffer(raw, dtype="<u8", count=m, offset=pos).copy()); pos += 8 * m
    return (fid, k, cont, val), out


def _read_edges(folder):
    raw = open(os.path.join(folder, "edge.mst"), "rb").read()
    (m,) = struct.unpack_from("<Q", raw, 0)
    return np.frombuffer(raw, dtype=np.dtype([("pre", "<i4"), ("suf", "<i4"), ("dist", "<f8")]), count=m, offset=8)


def test_clust_mst_end_to_end_and_resume(oracle, tmp_path):
    tmp = str(tmp_path)
    L = 2_000_000  # large enough that tune_parameters keeps -k 21
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 3, L, seed=4, two_records=True)
    out = os.path.join(tmp, "mst.out")
    mjson = os.path.join(tmp, "metrics.json")
    _run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "1000", "-d", "0.05", "-t", "4", "-o", out], tmp,
         env={"RTC_METRICS_JSON": mjson})
    folders = [d for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d)) and d[:2] == "20"]
    assert len(folders) == 1
    folder = os.path.join(tmp, folders[0])
    # the metrics file: the reference's phase labels and the sizes of the run (RTC_METRICS_JSON)
    import json
    m = json.load(open(mjson))
    assert m["command"] == "clust-mst" and m["sketch"] == "minhash" and m["genomes"] == 9 and m["kmer_size"] == 21
    assert m["bases"] == 9 * L and m["gpus"] >= 1
    assert all(m[key] > 0 for key in ("computing_sketch_s", "generateMST_s", "saveSketches_s", "saveMST_s", "total_s", "sketch_gbp_per_s"))
    # oracle on the same bytes (records of a file are separated, k-mers do not span them)
    parts, off = [], [0]
    for g, s in enumerate(seqs):
        b = s.tobytes()
        body = (b[:L // 3] + b"\n" + b[L // 3:]) if g % 2 else b
        parts.append(np.frombuffer(body, dtype=np.uint8)); off.append(off[-1] + len(body))
    want_sk = oracle.sketch_minhash_batch(np.concatenate(parts), np.array(off, dtype=np.uint64), 21, 1000)
    hdr, got_sk = _read_hash_sketch(folder)
    assert hdr == (0, 21, False, 1000)
    assert len(got_sk) == len(want_sk) and all(np.array_equal(a, b) for a, b in zip(got_sk, want_sk))
    flat, start, lens = oracle.to_csr(want_sk)
    want_mst = oracle.mst(flat, start, lens, 21, 0, 0.05)
    got_mst = _read_edges(folder)
    assert np.array_equal(np.sort(got_mst["dist"]).view(np.uint64), np.sort(want_mst["dist"]).view(np.uint64))
    want_cl = oracle.forest_clusters(want_mst, 0.05, len(seqs))
    got_cl = _parse_clusters(out)
    assert _partition(got_cl) == _partition(want_cl)
    assert m["clusters"] == len(got_cl) and m["mst_edges"] == len(got_mst)
    assert [c[0] for c in got_cl] == sorted(c[0] for c in got_cl)  # numbered by smallest member
    text = open(out).read()
    assert text.startswith("# Clustering threshold: 0.050000\n# Total clusters: %d\n#\n" % len(got_cl))
    assert ("\t%5d\t%6d\t%12dnt\t%20s\t%20s\t%s\n" % (0, 0, L, paths[0], "g0", "synthetic family 0")) in text
    # resume flows reproduce the same partition
    out2, out3 = os.path.join(tmp, "mst2.out"), os.path.join(tmp, "mst3.out")
    _run([os.path.join(BIN, "clust-mst"), "--presketched", folder, "-d", "0.05", "-o", out2], tmp)
    _run([os.path.join(BIN, "clust-mst"), "--premsted", folder, "-d", "0.05", "-o", out3], tmp)
    assert _partition(_parse_clusters(out2)) == _partition(got_cl) == _partition(_parse_clusters(out3))
    assert open(out3).read() == text  # --premsted re-cuts the saved MST: identical text


@pytest.mark.parametrize("shape", ["contigs", "gz"])
def test_clust_mst_real_input_shapes_match_the_oracle(oracle, tmp_path, shape):
    """What sketchFiles actually opens (src/SketchInfo.cpp:880-948): assemblies of ~200 contigs with N runs -- every record
    separator and N stretch is a run of the staging format, most waves of the packed sketch kernel still take the express walk --
    and gzip'd files (two of them with several members).  hash.sketch must equal the oracle run on the same records."""
    import gzip
    tmp = str(tmp_path)
    rng = np.random.default_rng(23)
    from rabbittclust_amd import api
    L = 2_400_000
    desc = api.synth_family_descs(2, 4, global_seed=61)
    paths, parts, off = [], [], [0]
    for g, d in enumerate(desc):
        a = oracle.synth_genome(int(d["fam_seed"]), int(d["mut_seed"]), int(d["mut_thr"]), L).copy()
        for _ in range(3):
            st = int(rng.integers(0, L - 600))
            a[st:st + int(rng.integers(1, 500))] = ord("N")
        cuts = [0]
        while cuts[-1] < L:
            cuts.append(min(L, cuts[-1] + (int(rng.integers(1000, 25_000)) if shape == "contigs" else L)))
        recs = [a[cuts[r]:cuts[r + 1]].tobytes() for r in range(len(cuts) - 1)]
        text = b"".join(f">g{g}_c{r} x\n".encode() + b"\n".join(rec[i:i + 80] for i in range(0, len(rec), 80)) + b"\n" for r, rec in enumerate(recs))
        pth = os.path.join(tmp, f"g{g}.fna" + (".gz" if shape == "gz" else ""))
        if shape == "gz":
            half = len(text) // 2 if g % 4 == 1 else len(text)  # some files as two gzip members
            with open(pth, "wb") as f:
                f.write(gzip.compress(text[:half], 6))
                if half < len(text):
                    f.write(gzip.compress(text[half:], 6))
        else:
            open(pth, "wb").write(text)
        paths.append(pth)
        body = b"\n".join(recs)  # records of a genome are separated: k-mers never span them
        parts.append(np.frombuffer(body, dtype=np.uint8))
        off.append(off[-1] + len(body))
    lst = os.path.join(tmp, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    out = os.path.join(tmp, "res.out")
    mj = os.path.join(tmp, "m.json")
    err = _run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "1000", "-d", "0.05", "-t", "4", "-o", out], tmp,
               env={"RTC_METRICS_JSON": mj, "RTC_VERBOSE": "1", "RTC_BATCH_BYTES": str(6 << 20)})
    assert "over packed bases" in err
    folder = [os.path.join(tmp, d) for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d))][0]
    want = oracle.sketch_minhash_batch(np.concatenate(parts), np.array(off, dtype=np.uint64), 21, 1000)
    _, got = _read_hash_sketch(folder)
    assert len(got) == len(want) and all(np.array_equal(x, y) for x, y in zip(got, want))
    import json
    m = json.load(open(mj))
    assert m["batches"] >= 2 and m["gpu_sketch_ms_per_batch"] > 0
    if shape == "contigs":
        assert m["runs_per_genome"] > 100
    else:
        assert m["inflate_gb_per_s_per_thread"] > 0


CLI_SOAK_SEEDS = int(os.environ.get("RTC_SOAK_SEEDS", "3"))


@pytest.mark.parametrize("seed", list(range(1, CLI_SOAK_SEEDS + 1)))
def test_cli_random_inputs_identical_across_stagings(oracle, tmp_path, seed):
    """Random small collections through both command lines' sketch paths: genomes of random length cut into random
    contigs, N runs and IUPAC codes of every length, lower case, CRLF, gzip (one or several members), tiny files below the
    length filter -- in batches of a few files.  The packed staging (default: the sketch kernels read the 2-bit stream)
    and the character staging (RTC_STAGE_ASCII=1) must write byte-identical hash.sketch / kssd.hash.sketch and cluster
    files; MinHash sketches are also compared with the oracle run on the same records."""
    import gzip
    tmp = str(tmp_path)
    rng = np.random.default_rng(7000 + seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    junk = np.frombuffer(b"NNNNnRYKMSWBDHV", dtype=np.uint8)
    paths, parts, off = [], [], [0]
    base = rng.choice(acgt, size=420_000)
    for g in range(int(rng.integers(8, 14))):
        L = int(rng.choice([20_000, 90_000, 150_000, 260_000, 400_000]))
        a = base[:L].copy()
        mut = rng.random(L) < rng.uniform(0.0, 0.06)
        a[mut] = rng.choice(acgt, size=int(mut.sum()))
        for _ in range(int(rng.integers(0, 6))):
            st = int(rng.integers(0, L))
            ln = int(rng.choice([1, 2, 7, 21, 64, 300, 5000]))
            a[st:st + ln] = rng.choice(junk, size=len(a[st:st + ln]))
        low = rng.random(L) < (0.3 if g % 3 == 0 else 0.0)
        a[low & (a > 64)] |= 0x20
        cuts = [0]
        while cuts[-1] < L:
            cuts.append(min(L, cuts[-1] + int(rng.choice([60, 1000, 9000, 40_000, L]))))
        recs = [a[cuts[r]:cuts[r + 1]].tobytes() for r in range(len(cuts) - 1)]
        width = int(rng.choice([60, 70, 80, 1 << 20]))
        eol = b"\r\n" if g % 4 == 1 else b"\n"
        text = b"".join(f">g{g}_c{r} d".encode() + eol + eol.join(rec[i:i + width] for i in range(0, len(rec), width)) + eol for r, rec in enumerate(recs))
        pth = os.path.join(tmp, f"g{g}.fna")
        if g % 4 == 2:
            pth += ".gz"
            half = len(text) // 3 if g % 8 == 2 else len(text)
            with open(pth, "wb") as f:
                f.write(gzip.compress(text[:half], 5))
                if half < len(text):
                    f.write(gzip.compress(text[half:], 5))
        else:
            open(pth, "wb").write(text)
        paths.append(pth)
        body = b"\n".join(recs)
        parts.append(np.frombuffer(body, dtype=np.uint8))
        off.append(off[-1] + len(body))
    lst = os.path.join(tmp, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    for tool, extra, sk_name in (("clust-mst", ["-s", "400"], "hash.sketch"), ("clust-mst", ["--fast"], "kssd.hash.sketch"),
                                 ("clust-greedy", ["-c", "500"], "hash.sketch")):
        outs = {}
        for tag, env in (("packed", {}), ("ascii", {"RTC_STAGE_ASCII": "1"})):
            d = os.path.join(tmp, tool + extra[0].strip("-") + tag)
            os.makedirs(d)
            out = os.path.join(d, "res.out")
            _run([os.path.join(BIN, tool), "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "-m", "10000", "-o", out] + extra, d,
                 env=dict(env, RTC_BATCH_BYTES=str(1 << 20)))
            folder = [os.path.join(d, x) for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))][0]
            outs[tag] = (open(out).read().replace(d, ""), open(os.path.join(folder, sk_name), "rb").read(), folder)
        assert outs["packed"][1] == outs["ascii"][1], (tool, extra, "sketch files differ between the stagings")
        assert outs["packed"][0] == outs["ascii"][0], (tool, extra, "cluster files differ between the stagings")
        if tool == "clust-mst" and extra == ["-s", "400"]:
            hdr, got = _read_hash_sketch(outs["packed"][2])  # (tune_parameters may have lowered k for genomes this small: the header says)
            want = oracle.sketch_minhash_batch(np.concatenate(parts), np.array(off, dtype=np.uint64), hdr[1], 400)
            assert len(got) == len(want) and all(np.array_equal(x, y) for x, y in zip(got, want))


def test_clust_mst_fast_kssd_end_to_end(oracle, tmp_path):
    tmp = str(tmp_path)
    L = 2_000_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 2, 3, L, seed=5)
    out = os.path.join(tmp, "kssd.out")
    _run([os.path.join(BIN, "clust-mst"), "--fast", "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "-o", out], tmp)
    want_sk = [oracle.kssd_sketch(s, 21, 3) for s in seqs]
    folder = [os.path.join(tmp, d) for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d)) and d[:2] == "20"][0]
    raw = open(os.path.join(folder, "kssd.hash.sketch"), "rb").read()
    assert struct.unpack_from("<iiiii", raw, 0) == ((11 << 8) + (6 << 4) + 3, 11, 6, 3, len(seqs))
    pos = 20
    for w in want_sk:
        (m,) = struct.unpack_from("<Q", raw, pos); pos += 8
        assert np.array_equal(np.frombuffer(raw, dtype="<u4", count=m, offset=pos), w); pos += 4 * m
    flat, start, lens = oracle.to_csr(want_sk, dtype=np.uint32)
    want_mst = oracle.mst(flat, start, lens, 22, 0, 0.05)
    assert _partition(_parse_clusters(out)) == _partition(oracle.forest_clusters(want_mst, 0.05, len(seqs)))
    for f in ("kssd.info.sketch", "kssd.sketch.index", "kssd.sketch.dict", "kssd.info.mst", "edge.mst"):
        assert os.path.exists(os.path.join(folder, f)), f


def test_clust_greedy_default_containment_end_to_end(oracle, tmp_path):
    tmp = str(tmp_path)
    L = 2_000_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 3, L, seed=6)
    out = os.path.join(tmp, "greedy.out")
    _run([os.path.join(BIN, "clust-greedy"), "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "-e", "-o", out], tmp)
    sizes = [os.path.getsize(p) for p in paths]
    compress = (sum(sizes) // len(sizes)) // 1000           # tune_parameters GREEDY default
    cfg = np.array([max(s // compress, 100) for s in sizes], dtype=np.uint32)
    off = np.arange(len(seqs) + 1, dtype=np.uint64) * L
    sk = oracle.sketch_minhash_batch(np.concatenate(seqs), off, 21, cfg)
    flat, start, lens = oracle.to_csr(sk)
    ncl, rep = oracle.greedy_minhash(flat, start, lens, cfg, 21, True, 0.05)
    got = _parse_clusters(out)
    assert len(got) == ncl
    want = {}
    for i, r in enumerate(rep):
        want.setdefault(int(r), []).append(i)
    assert [c for c in got] == [[r] + [m for m in ms if m != r] for r, ms in sorted(want.items())]
    assert not open(out).read().startswith("#")            # greedy output has no threshold header
    assert not [d for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d)) and d[:2] == "20"]  # -e


def test_gpu_sketches_match_committed_fixture(ctx):
    fx = np.load(os.path.join(ROOT, "tests", "golden", "sketch_fixture.npz"), allow_pickle=True)
    from rabbittclust_amd import api
    L = int(fx["L"])
    desc = np.zeros(len(fx["descs"]), dtype=api.SYNTH_DT)
    for i, (f, m, t, ne) in enumerate(fx["descs"]):
        desc[i] = (int(f), int(m), int(t), int(ne))
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    mh = ctx.sketch_minhash(seq, off, k=21, size=400).to_host()
    assert all(np.array_equal(a, b) for a, b in zip(mh, fx["minhash"]))
    import ctypes as C
    host = C.CDLL(os.path.join(ROOT, "rabbittclust_amd", "librtclust_host.so"))
    sd = np.zeros(1 << 24, dtype=np.int32)
    host.rtch_shuffle_dim(6, sd.ctypes.data_as(C.c_void_p))
    ks = ctx.sketch_kssd(seq, off, sd, kmer_size=21, drlevel=3).to_host()
    assert all(np.array_equal(a, b) for a, b in zip(ks, fx["kssd"]))
    sk = ctx.sketch_minhash(seq, off, k=21, size=400)
    mst = ctx.mst(sk, 0.05)
    assert np.array_equal(np.sort(mst["dist"]).view(np.uint64), np.sort(fx["mst"]["dist"]).view(np.uint64))


def test_clust_greedy_fast_and_presketched(oracle, tmp_path):
    """clust-greedy --fast (KSSD, size-sorted) and clust-greedy --presketched on a MinHash folder
    written by clust-mst (fixed-size fast path, genomes re-sorted by length)."""
    tmp = str(tmp_path)
    L = 2_000_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 3, L, seed=8)
    out = os.path.join(tmp, "gfast.out")
    _run([os.path.join(BIN, "clust-greedy"), "--fast", "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "-e", "-o", out], tmp)
    ks = [oracle.kssd_sketch(s, 21, 3) for s in seqs]
    # src/greedy.cpp:594-597: std::sort by hash count, descending (same libstdc++ sort on both sides)
    import ctypes as C
    order = sorted(range(len(ks)), key=lambda i: -len(ks[i]))
    sizes = [len(ks[i]) for i in order]
    assert len(set(sizes)) == len(sizes), "test data must not tie on sketch size (unstable sort order)"
    flat, start, lens = oracle.to_csr([ks[i] for i in order], dtype=np.uint32)
    ncl, rep = oracle.greedy_kssd(flat, start, lens, 22, 0.05)
    got = _parse_clusters(out)
    assert len(got) == ncl
    # ids in the output are positions in the size-sorted order; compare as sets of original genomes
    want = {}
    for i, r in enumerate(rep):
        want.setdefault(int(r), []).append(i)
    assert _partition(got) == _partition(list(want.values()))
    # the file names printed for each id identify the original genome
    names = {}
    for ln in open(out):
        if ln.startswith("\t"):
            f = ln.rstrip("\n").split("\t")
            names[int(f[2])] = f[4].strip()
    assert [names[i] for i in range(len(order))] == [paths[i] for i in order]

    # MinHash folder from clust-mst, then clust-greedy --presketched
    out_m = os.path.join(tmp, "m.out")
    _run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "1000", "-d", "0.05", "-t", "4", "-o", out_m], tmp)
    folder = [os.path.join(tmp, d) for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d)) and d[:2] == "20"][0]
    out_g = os.path.join(tmp, "g2.out")
    _run([os.path.join(BIN, "clust-greedy"), "--presketched", folder, "-d", "0.05", "-o", out_g], tmp)
    off = np.arange(len(seqs) + 1, dtype=np.uint64) * L
    sk = oracle.sketch_minhash_batch(np.concatenate(seqs), off, 21, 1000)
    flat, start, lens = oracle.to_csr(sk)   # all genomes have the same length: cmpGenomeSize keeps id order
    ncl, rep = oracle.greedy_minhash(flat, start, lens, 1000, 21, False, 0.05)
    got = _parse_clusters(out_g)
    want = {}
    for i, r in enumerate(rep):
        want.setdefault(int(r), []).append(i)
    assert got == [[r] + [m for m in ms if m != r] for r, ms in sorted(want.items())]


def _kssd_dist(cm, sr, sq, kmer):
    jac = cm / (sq + sr - cm)
    return 0.0 if jac == 1.0 else min(1.0, -math.log(2 * jac / (1.0 + jac)) / kmer)


def _rep_candidates(q, rep_sets, kmer, thr):
    """(distance, representative position) of every representative that survives the walk and the filters of
    KssdIncrementalCluster / query_topk (src/greedy.cpp:1768-1835, :2545-2600)."""
    radio = 2.0 * math.exp(thr * kmer) - 1.0
    x = math.exp(-thr * kmer); jmin = x / (2.0 - x)
    out = []
    for r, rs in enumerate(rep_sets):
        cm = len(q & rs)
        if cm == 0:
            continue
        sq, sr = len(q), len(rs)
        ratio = sq / sr
        if ratio > radio or ratio < 1.0 / radio:
            continue
        if cm < int(jmin * (sq + sr) / (1.0 + jmin)):
            continue
        out.append((_kssd_dist(cm, sr, sq, kmer), r))
    return sorted(out)


def _incremental_twin(sets, n_old, cid, clusters, kmer, thr):
    """KssdIncrementalCluster (src/greedy.cpp:1736-1900) over sets[n_old:]; cid: representative genome -> cluster."""
    for q in range(n_old, len(sets)):
        reps = sorted(cid, key=lambda g: cid[g])
        cand = [c for c in _rep_candidates(sets[q], [sets[g] for g in reps], kmer, thr) if c[0] <= thr]
        if cand:
            clusters[cand[0][1]].append(q)
        else:
            cid[q] = len(clusters); clusters.append([])


def test_clust_greedy_fast_append(oracle, tmp_path):
    """clust-greedy --fast --presketched DIR --append LIST without a stored cluster state
    (append_clust_greedy_fast, "Initial State Building Mode"): the stored sketches are clustered as
    clust-greedy --fast clusters them (size-sorted), then every new genome in input order joins the nearest
    representative that passes KssdIncrementalCluster's filters (src/greedy.cpp:1736-1900) or opens a cluster.
    Checked against a Python restatement on the oracle's KSSD sketches."""
    import math
    tmp = str(tmp_path)
    L = 2_000_000  # (shorter genomes make tune_parameters replace -k 21, src/sub_command.cpp:2414-2430)
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 4, 4, L, seed=34)
    perm = [0, 5, 10, 1, 4, 8, 9, 2, 3, 6, 7, 13, 11, 12, 14, 15]  # family 3 arrives with the appended genomes only
    first, second = perm[:7], perm[7:]
    la, lb = os.path.join(tmp, "a.txt"), os.path.join(tmp, "b.txt")
    open(la, "w").write("\n".join(paths[i] for i in first) + "\n")
    open(lb, "w").write("\n".join(paths[i] for i in second) + "\n")
    da = os.path.join(tmp, "a"); os.makedirs(da)
    _run([os.path.join(BIN, "clust-greedy"), "--fast", "-l", "-i", la, "-k", "21", "-d", "0.05", "-t", "4", "-o", os.path.join(da, "a.out")], da)
    folder = [os.path.join(da, d) for d in os.listdir(da) if os.path.isdir(os.path.join(da, d))][0]
    out = os.path.join(tmp, "ab.out")
    err = _run([os.path.join(BIN, "clust-greedy"), "--fast", "-l", "--presketched", folder, "--append", lb, "-d", "0.05", "-t", "4", "-e",
                "-o", out], tmp)
    assert "Initial State Building Mode (KSSD)" in err and "---the half_k is: 11" in err
    # ---- restatement ----
    ks = {i: oracle.kssd_sketch(seqs[i], 21, 3) for i in perm}
    pre = sorted(first, key=lambda i: -len(ks[i]))
    assert len({len(ks[i]) for i in pre}) == len(pre), "test data must not tie on sketch size (unstable sort order)"
    order = pre + second
    flat, start, lens = oracle.to_csr([ks[i] for i in pre], dtype=np.uint32)
    ncl, rep = oracle.greedy_kssd(flat, start, lens, 22, 0.05)
    clusters, cid = [], {}
    for pos, r in enumerate(rep):
        if int(r) == pos:
            cid[pos] = len(clusters); clusters.append([pos])
    for pos, r in enumerate(rep):
        if int(r) != pos:
            clusters[cid[int(r)]].append(pos)
    clusters_pre = [list(c) for c in clusters]
    sets = [set(ks[i].tolist()) for i in order]
    _incremental_twin(sets, len(pre), cid, clusters, 22, 0.05)
    got = _parse_clusters(out)
    assert got == clusters
    # the appended list opened clusters of its own; a genome that opens one is its representative but is not listed
    # among the members (src/greedy.cpp:1861-1864)
    new_reps = [g for g in cid if g >= len(pre)]
    assert len(clusters) < 16 and new_reps and any(clusters[cid[g]] for g in new_reps)
    listed = sorted(m for c in clusters for m in c)
    assert listed == [g for g in range(len(order)) if g not in new_reps]
    names = {}
    for ln in open(out):
        if ln.startswith("\t"):
            f = ln.rstrip("\n").split("\t")
            names[int(f[2])] = f[4].strip()
    assert [names[i] for i in listed] == [paths[order[i]] for i in listed]
    # ---- the same through a stored cluster state: --save-rep writes DIR/cluster_state.bin, --append finds it ----
    import struct
    ds_ = os.path.join(tmp, "s"); os.makedirs(ds_)
    _run([os.path.join(BIN, "clust-greedy"), "--fast", "-l", "-i", la, "-k", "21", "-d", "0.05", "-t", "4", "--save-rep", "-o",
          os.path.join(ds_, "a.out")], ds_)
    sfolder = [os.path.join(ds_, d) for d in os.listdir(ds_) if os.path.isdir(os.path.join(ds_, d))][0]
    state = os.path.join(sfolder, "cluster_state.bin")

    def read_state(path):  # KssdClusterState::save, src/greedy.cpp:1545-1625
        b = open(path, "rb").read()
        thr_, k_, hk, hs, dr, gn = struct.unpack_from("<diiiii", b, 0)
        p_ = 28
        (nrep,) = struct.unpack_from("<Q", b, p_); p_ += 8
        reps = list(struct.unpack_from("<%di" % nrep, b, p_)); p_ += 4 * nrep
        (nsk,) = struct.unpack_from("<Q", b, p_); p_ += 8
        names_, sks = [], []
        for _ in range(nsk):
            gid, tot, u64, ssz = struct.unpack_from("<iQ?I", b, p_); p_ += 4 + 8 + 1 + 4
            n32, n64 = struct.unpack_from("<QQ", b, p_); p_ += 16
            sks.append(np.frombuffer(b, dtype=np.uint32, count=n32, offset=p_).copy()); p_ += 4 * n32 + 8 * n64
            (nl,) = struct.unpack_from("<Q", b, p_); p_ += 8
            names_.append(b[p_:p_ + nl].decode()); p_ += nl
            assert ssz == n32 and n64 == 0 and not u64
        (ncl_,) = struct.unpack_from("<Q", b, p_); p_ += 8
        cls = []
        for _ in range(ncl_):
            (m,) = struct.unpack_from("<Q", b, p_); p_ += 8
            cls.append(list(struct.unpack_from("<%di" % m, b, p_))); p_ += 4 * m
        assert b[p_:p_ + 8] == b"KSSI02\0\0"; p_ += 8
        (nidx,) = struct.unpack_from("<Q", b, p_); p_ += 8
        index = {}
        for _ in range(nidx):
            h, ls = struct.unpack_from("<QQ", b, p_); p_ += 16
            index[h] = list(struct.unpack_from("<%di" % ls, b, p_)); p_ += 4 * ls
        assert p_ == len(b)
        return dict(thr=thr_, k=k_, half_k=hk, drlevel=dr, n=gn, reps=reps, names=names_, sk=sks, clusters=cls, index=index)

    st = read_state(state)
    pre_clusters = [c for c in (clusters_pre)]
    assert (st["thr"], st["k"], st["half_k"], st["drlevel"], st["n"]) == (0.05, 22, 11, 3, len(pre))
    assert st["clusters"] == pre_clusters and st["reps"] == [c[0] for c in pre_clusters]
    assert st["names"] == [paths[i] for i in pre] and all(np.array_equal(a, ks[i]) for a, i in zip(st["sk"], pre))
    want_index = {}
    for ridx, g in enumerate(st["reps"]):
        for h in ks[pre[g]].tolist():
            want_index.setdefault(h, []).append(ridx)
    assert st["index"] == want_index
    half = len(second) // 2
    lb1, lb2 = os.path.join(tmp, "b1.txt"), os.path.join(tmp, "b2.txt")
    open(lb1, "w").write("\n".join(paths[i] for i in second[:half]) + "\n")
    open(lb2, "w").write("\n".join(paths[i] for i in second[half:]) + "\n")
    out1, out2 = os.path.join(tmp, "s1.out"), os.path.join(tmp, "s2.out")
    d1 = os.path.join(tmp, "s1"); os.makedirs(d1)
    err = _run([os.path.join(BIN, "clust-greedy"), "--fast", "-l", "--presketched", sfolder, "--append", lb1, "-d", "0.2", "-t", "4",
                "--save-rep", "-o", out1], d1)
    assert "Incremental Update Mode (KSSD)" in err  # (-d 0.2 is ignored: the state's threshold decides)
    st1 = read_state(state)
    assert st1["n"] == len(pre) + half and st1["names"] == [paths[i] for i in order[:len(pre) + half]]
    reps_now = sorted((g for g in cid if g < len(pre) + half), key=lambda g: cid[g])
    assert st1["reps"] == reps_now and len(st1["clusters"]) == len(reps_now)
    assert st1["clusters"] == [[m for m in clusters[cid[g]] if m < len(pre) + half] for g in reps_now]
    err = _run([os.path.join(BIN, "clust-greedy"), "--fast", "-l", "--presketched", sfolder, "--append", lb2, "-d", "0.05", "-t", "4", "-e",
                "-o", out2], tmp)
    assert "Incremental Update Mode (KSSD)" in err
    assert _parse_clusters(out2) == clusters          # two appends through the state = one stateless append of everything
    assert read_state(state)["n"] == len(pre) + half  # -e: the state stays as it was
    seqnames = {}
    for ln in open(out2):
        if ln.startswith("\t"):
            f = ln.rstrip("\n").split("\t")
            seqnames[int(f[2])] = (f[4].strip(), f[5].strip())
    assert all(seqnames[i][1] == "N/A" for i in listed if i < len(pre) + half)          # src/MST_IO.cpp:99-104
    assert all(seqnames[i][1] != "N/A" for i in listed if i >= len(pre) + half)
    assert [seqnames[i][0] for i in listed] == [paths[order[i]] for i in listed]
    # usage errors (src/main.cpp:378-381); without --fast the MinHash flow looks for hash.sketch in the KSSD folder
    r = subprocess.run([os.path.join(BIN, "clust-greedy"), "--fast", "-l", "--append", lb, "-o", out], capture_output=True, text=True)
    assert r.returncode != 0 and "--presketched needed" in r.stderr
    r = subprocess.run([os.path.join(BIN, "clust-greedy"), "-l", "--presketched", folder, "--append", lb, "-o", out], capture_output=True, text=True)
    assert r.returncode != 0 and "hash.sketch" in r.stderr


def _read_repdb(path):
    """KssdClusterState::save_repdb (src/greedy.cpp:2351-2428), field by field."""
    import struct
    b = open(path, "rb").read()
    assert b[:8] == b"REPDB002"
    thr, k, hk, hs, dr, gn = struct.unpack_from("<diiiii", b, 8)
    p = 36
    (nrep,) = struct.unpack_from("<Q", b, p); p += 8
    rep_ids, rep_names, rep_len, rep_sk = [], [], [], []
    for _ in range(nrep):
        rid, gid, tot, u64, ssz = struct.unpack_from("<iiQ?I", b, p); p += 4 + 4 + 8 + 1 + 4
        n32, n64 = struct.unpack_from("<QQ", b, p); p += 16
        assert not u64 and n64 == 0 and ssz == n32
        rep_sk.append(np.frombuffer(b, dtype=np.uint32, count=n32, offset=p).copy()); p += 4 * n32
        (nl,) = struct.unpack_from("<Q", b, p); p += 8
        rep_names.append(b[p:p + nl].decode()); p += nl
        rep_ids.append(rid); rep_len.append(tot)
    (ncl,) = struct.unpack_from("<Q", b, p); p += 8
    cls = []
    for _ in range(ncl):
        (m,) = struct.unpack_from("<Q", b, p); p += 8
        cls.append(list(struct.unpack_from("<%di" % m, b, p))); p += 4 * m
    (nall,) = struct.unpack_from("<Q", b, p); p += 8
    names, lens = [], []
    for _ in range(nall):
        (nl,) = struct.unpack_from("<Q", b, p); p += 8
        names.append(b[p:p + nl].decode()); p += nl
        (tot,) = struct.unpack_from("<Q", b, p); p += 8
        lens.append(tot)
    (nidx,) = struct.unpack_from("<Q", b, p); p += 8
    index = {}
    for _ in range(nidx):
        h, ls = struct.unpack_from("<QQ", b, p); p += 16
        index[h] = list(struct.unpack_from("<%di" % ls, b, p)); p += 4 * ls
    assert p == len(b)
    return dict(thr=thr, k=k, half_k=hk, half_subk=hs, drlevel=dr, n=gn, rep_ids=rep_ids, rep_names=rep_names, rep_len=rep_len,
                rep_sk=rep_sk, clusters=cls, names=names, lens=lens, index=index)


def test_clust_greedy_fast_repdb(oracle, tmp_path):
    """clust-greedy --fast --db FILE: --build (from a sketch folder and from genomes), --stats, --query --top-k,
    --assign and --append (src/sub_command.cpp:276-496, src/greedy.cpp:2351-2765) against a Python restatement on
    the oracle's KSSD sketches; the RepDB file is parsed field by field."""
    tmp = str(tmp_path)
    L = 2_000_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 4, 4, L, seed=34)
    perm = [0, 5, 10, 1, 4, 8, 9, 2, 3, 6, 7, 13, 11, 12, 14, 15]  # family 3 is absent from the database
    first, second = perm[:7], perm[7:]
    la, lb = os.path.join(tmp, "a.txt"), os.path.join(tmp, "b.txt")
    open(la, "w").write("\n".join(paths[i] for i in first) + "\n")
    open(lb, "w").write("\n".join(paths[i] for i in second) + "\n")
    G = os.path.join(BIN, "clust-greedy")
    da = os.path.join(tmp, "a"); os.makedirs(da)
    db, db2 = os.path.join(tmp, "rep.db"), os.path.join(tmp, "rep2.db")
    # --build from genomes: sketches with -k as given (no tuning), writes the sketch folder, clusters, saves the RepDB
    err = _run([G, "--fast", "--db", db, "--build", "-l", "-i", la, "-k", "21", "-d", "0.05", "-t", "4", "-o", os.path.join(da, "a.out")], da)
    assert "===== RepDB Build (from genomes) =====" in err and "RepDB saved to: " + db in err
    folder = [os.path.join(da, d) for d in os.listdir(da) if os.path.isdir(os.path.join(da, d))][0]
    # --build from the folder: the same database
    err = _run([G, "--fast", "--db", db2, "--build", "--presketched", folder, "-d", "0.05", "-o", os.path.join(da, "a2.out")], da)
    assert "===== RepDB Build (from pre-sketched) =====" in err
    assert open(db, "rb").read() == open(db2, "rb").read()
    assert open(os.path.join(da, "a.out")).read() == open(os.path.join(da, "a2.out")).read()
    assert open(os.path.join(da, "a.out")).read().startswith("# Clustering threshold: 0.050000\n# Total clusters: ")
    # ---- restatement of the build ----
    ks = {i: oracle.kssd_sketch(seqs[i], 21, 3) for i in perm}
    pre = sorted(first, key=lambda i: -len(ks[i]))
    assert len({len(ks[i]) for i in pre}) == len(pre)
    flat, start, lens = oracle.to_csr([ks[i] for i in pre], dtype=np.uint32)
    ncl, rep = oracle.greedy_kssd(flat, start, lens, 22, 0.05)
    clusters, cid = [], {}
    for pos, r in enumerate(rep):
        if int(r) == pos:
            cid[pos] = len(clusters); clusters.append([pos])
    for pos, r in enumerate(rep):
        if int(r) != pos:
            clusters[cid[int(r)]].append(pos)
    assert _parse_clusters(os.path.join(da, "a.out")) == clusters
    d = _read_repdb(db)
    reps = sorted(cid, key=lambda g: cid[g])
    assert (d["thr"], d["k"], d["half_k"], d["drlevel"]) == (0.05, 22, 11, 3)
    assert d["rep_ids"] == reps and d["clusters"] == clusters
    assert d["rep_names"] == [paths[pre[g]] for g in reps] and d["names"] == [paths[i] for i in pre]
    assert d["lens"] == [L] * len(pre) and d["rep_len"] == [L] * len(reps)
    assert all(np.array_equal(a, ks[pre[g]]) for a, g in zip(d["rep_sk"], reps))
    want_index = {}
    for ridx, g in enumerate(reps):
        for h in ks[pre[g]].tolist():
            want_index.setdefault(h, []).append(ridx)
    assert d["index"] == want_index
    # ---- --stats ----
    r = subprocess.run([G, "--fast", "--db", db, "--stats"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = r.stdout
    assert "        RepDB Statistics Report" in txt
    assert "  Total genomes:          %d\n" % len(pre) in txt and "  Representatives:        %d\n" % len(reps) in txt
    assert "  Compression ratio:      %.2f%%\n" % ((1.0 - len(reps) / len(pre)) * 100.0) in txt
    assert "  Unique hashes:          %d\n" % len(want_index) in txt
    assert "  Total postings:         %d\n" % sum(len(v) for v in want_index.values()) in txt
    assert "  Max posting length:     %d\n" % max(len(v) for v in want_index.values()) in txt
    sizes = sorted(len(c) for c in clusters)
    assert "  Median cluster size:    %d\n" % sizes[len(sizes) // 2] in txt
    assert "  Max sketch size:        %d\n" % max(len(ks[pre[g]]) for g in reps) in txt
    assert "  Total sequence length:  %d bp\n" % (L * len(pre)) in txt and "  Coverage ratio:         %.2f%%\n" % (100.0 * len(reps) / len(pre)) in txt
    # ---- --query / --assign ----
    rep_sets = [set(ks[pre[g]].tolist()) for g in reps]
    qout, aout = os.path.join(tmp, "q.tsv"), os.path.join(tmp, "as.tsv")
    _run([G, "--fast", "--db", db, "--query", "-l", "-i", lb, "--top-k", "2", "-t", "4", "-o", qout], tmp)
    _run([G, "--fast", "--db", db, "--assign", "-l", "-i", lb, "-t", "4", "-o", aout], tmp)
    want_q = ["#query\trank\trep_name\tdistance\tcluster_id\tcluster_size\n"]
    want_a = ["#query\tassigned_cluster\trep_name\tdistance\tcluster_size\tstatus\n"]
    n_match = n_none = 0
    for i in second:
        cand = _rep_candidates(set(ks[i].tolist()), rep_sets, 22, 0.05)
        assert len({c[0] for c in cand}) == len(cand)  # no distance ties in the test data
        if not cand:
            want_q.append("%s\t0\tno_match\t-1\t-1\t0\n" % paths[i]); n_none += 1
        for rank, (dist, r) in enumerate(cand[:2]):
            want_q.append("%s\t%d\t%s\t%.6f\t%d\t%d\n" % (paths[i], rank + 1, paths[pre[reps[r]]], dist, r, len(clusters[r])))
        if cand and cand[0][0] <= 0.05:
            want_a.append("%s\t%d\t%s\t%.6f\t%d\tassigned\n" % (paths[i], cand[0][1], paths[pre[reps[cand[0][1]]]], cand[0][0], len(clusters[cand[0][1]])))
            n_match += 1
        else:
            want_a.append("%s\t-1\tunassigned\t-1\t0\tnovel\n" % paths[i])
    assert n_match >= 4 and n_none >= 3
    assert open(qout).readlines() == want_q
    assert open(aout).readlines() == want_a
    assert open(db, "rb").read() == open(db2, "rb").read()  # read-only actions
    # ---- --append: KssdIncrementalCluster on the database, which is rewritten ----
    apout = os.path.join(tmp, "ap.out")
    err = _run([G, "--fast", "--db", db, "--append", lb, "-l", "-t", "4", "-o", apout], tmp)
    assert "===== RepDB Append =====" in err and "  RepDB updated:    " + db in err
    order = pre + second
    sets = [set(ks[i].tolist()) for i in order]
    _incremental_twin(sets, len(pre), cid, clusters, 22, 0.05)
    assert _parse_clusters(apout) == clusters
    assert open(apout).read().startswith("# Clustering threshold: 0.050000\n# Total clusters: %d\n#\n" % len(clusters))
    d = _read_repdb(db)
    reps = sorted(cid, key=lambda g: cid[g])
    assert d["rep_ids"] == reps and d["clusters"] == clusters and d["n"] == len(order)
    assert d["names"] == [paths[i] for i in order] and d["rep_names"] == [paths[order[g]] for g in reps]
    assert all(np.array_equal(a, ks[order[g]]) for a, g in zip(d["rep_sk"], reps))
    # a second append works from the stored representatives alone (the RepDB keeps no other sketches)
    lc = os.path.join(tmp, "c.txt")
    open(lc, "w").write("\n".join(paths[i] for i in (13, 2)) + "\n")
    _run([G, "--fast", "--db", db, "--append", lc, "-l", "-t", "4", "-o", apout], tmp)
    sets += [set(ks[13].tolist()), set(ks[2].tolist())]
    _incremental_twin(sets, len(order), cid, clusters, 22, 0.05)
    assert _parse_clusters(apout) == clusters and cid == {g: c for c, g in enumerate(reps)}  # both joined existing clusters
    # usage errors
    r = subprocess.run([G, "--fast", "--db", db, "-o", apout], capture_output=True, text=True)
    assert r.returncode != 0 and "--db requires one of: --build, --query, --assign, --append, --stats" in r.stderr
    r = subprocess.run([G, "--fast", "--db", db, "--query", "-o", apout], capture_output=True, text=True)
    assert r.returncode != 0 and "--query requires -i <input_file>" in r.stderr
    r = subprocess.run([G, "--db", db, "--stats"], capture_output=True, text=True)  # without --fast: read as a MinHash RepDB
    assert r.returncode != 0 and "Invalid MinHash RepDB file (bad magic)" in r.stderr


def _read_mh_repdb(path):
    """MinHashClusterState::save_repdb (src/greedy.cpp:2789-2862), field by field."""
    import struct
    b = open(path, "rb").read()
    assert b[:8] == b"MHREPDB1"
    thr, k, ssz, cont = struct.unpack_from("<dii?", b, 8)
    p = 8 + 8 + 4 + 4 + 1
    (nrep,) = struct.unpack_from("<Q", b, p); p += 8
    rep_ids, rep_names, rep_len, rep_sk = [], [], [], []
    for _ in range(nrep):
        rid, gid, tot, c2 = struct.unpack_from("<iiQ?", b, p); p += 4 + 4 + 8 + 1
        assert c2 == cont
        (nh,) = struct.unpack_from("<Q", b, p); p += 8
        rep_sk.append(np.frombuffer(b, dtype=np.uint64, count=nh, offset=p).copy()); p += 8 * nh
        (nl,) = struct.unpack_from("<Q", b, p); p += 8
        rep_names.append(b[p:p + nl].decode()); p += nl
        rep_ids.append(rid); rep_len.append(tot)
    (ncl,) = struct.unpack_from("<Q", b, p); p += 8
    cls = []
    for _ in range(ncl):
        (m,) = struct.unpack_from("<Q", b, p); p += 8
        cls.append(list(struct.unpack_from("<%di" % m, b, p))); p += 4 * m
    (nall,) = struct.unpack_from("<Q", b, p); p += 8
    names, lens = [], []
    for _ in range(nall):
        (nl,) = struct.unpack_from("<Q", b, p); p += 8
        names.append(b[p:p + nl].decode()); p += nl
        (tot,) = struct.unpack_from("<Q", b, p); p += 8
        lens.append(tot)
    (nidx,) = struct.unpack_from("<Q", b, p); p += 8
    index = {}
    for _ in range(nidx):
        h, ls = struct.unpack_from("<QQ", b, p); p += 16
        index[h] = list(struct.unpack_from("<%di" % ls, b, p)); p += 4 * ls
    assert p == len(b)
    return dict(thr=thr, k=k, sketch_size=ssz, cont=cont, rep_ids=rep_ids, rep_names=rep_names, rep_len=rep_len, rep_sk=rep_sk,
                clusters=cls, names=names, lens=lens, index=index)


def _mh_dist(cm, sq, sr, kmer):
    """minhash_mash_distance, fixed-size mode (src/greedy.cpp:2771-2787)."""
    if cm <= 0:
        return 1.0
    den = sq + sr - cm
    if den == 0:
        return 0.0
    jac = cm / den
    if jac >= 1.0:
        return 0.0
    return min(1.0, -math.log(2.0 * jac / (1.0 + jac)) / kmer)


def test_clust_greedy_minhash_repdb(oracle, tmp_path):
    """clust-greedy --db FILE on MinHash sketches (mh_repdb_*, src/sub_command.cpp:502-758; MinHashClusterState,
    src/greedy.cpp:1903-2130, :2771-3147): --build from genomes and from a folder (list order, no size sort), --stats,
    --query, --assign, --append, against a Python restatement on the oracle's sketches."""
    tmp = str(tmp_path)
    L = 2_000_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 4, 4, L, seed=34)
    perm = [0, 5, 10, 1, 4, 8, 9, 2, 3, 6, 7, 13, 11, 12, 14, 15]  # family 3 is absent from the database
    first, second = perm[:7], perm[7:]
    la, lb = os.path.join(tmp, "a.txt"), os.path.join(tmp, "b.txt")
    open(la, "w").write("\n".join(paths[i] for i in first) + "\n")
    open(lb, "w").write("\n".join(paths[i] for i in second) + "\n")
    G = os.path.join(BIN, "clust-greedy")
    da = os.path.join(tmp, "a"); os.makedirs(da)
    db, db2 = os.path.join(tmp, "rep.db"), os.path.join(tmp, "rep2.db")
    err = _run([G, "--db", db, "--build", "-l", "-i", la, "-k", "21", "-s", "1000", "-d", "0.05", "-t", "4", "-o", os.path.join(da, "a.out")], da)
    assert "===== MinHash RepDB Build (from genomes) =====" in err and "MinHash RepDB saved to: " + db in err
    folder = [os.path.join(da, d) for d in os.listdir(da) if os.path.isdir(os.path.join(da, d))][0]
    err = _run([G, "--db", db2, "--build", "--presketched", folder, "-d", "0.05", "-o", os.path.join(da, "a2.out")], da)
    assert "===== MinHash RepDB Build (from pre-sketched) =====" in err
    assert open(db, "rb").read() == open(db2, "rb").read()
    assert open(os.path.join(da, "a.out")).read() == open(os.path.join(da, "a2.out")).read()
    # ---- restatement of the build: the MinHash greedy pass in list order ----
    off = np.arange(len(seqs) + 1, dtype=np.uint64) * L
    sk_all = oracle.sketch_minhash_batch(np.concatenate(seqs), off, 21, 1000)
    sk = {i: sk_all[i] for i in perm}
    flat, start, lens = oracle.to_csr([sk[i] for i in first])
    ncl, rep = oracle.greedy_minhash(flat, start, lens, 1000, 21, False, 0.05)
    clusters, cid = [], {}
    for pos, r in enumerate(rep):
        if int(r) == pos:
            cid[pos] = len(clusters); clusters.append([pos])
    for pos, r in enumerate(rep):
        if int(r) != pos:
            clusters[cid[int(r)]].append(pos)
    assert _parse_clusters(os.path.join(da, "a.out")) == clusters and 1 < len(clusters) < len(first)
    d = _read_mh_repdb(db)
    reps = sorted(cid, key=lambda g: cid[g])
    assert (d["thr"], d["k"], d["sketch_size"], d["cont"]) == (0.05, 21, 1000, False)
    assert d["rep_ids"] == reps and d["clusters"] == clusters
    assert d["rep_names"] == [paths[first[g]] for g in reps] and d["names"] == [paths[i] for i in first]
    assert d["lens"] == [L] * len(first) and d["rep_len"] == [L] * len(reps)
    assert all(np.array_equal(a, sk[first[g]]) for a, g in zip(d["rep_sk"], reps))
    want_index = {}
    for ridx, g in enumerate(reps):
        for h in sk[first[g]].tolist():
            want_index.setdefault(h, []).append(ridx)
    assert d["index"] == want_index
    # ---- --stats ----
    r = subprocess.run([G, "--db", db, "--stats"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = r.stdout
    assert "    MinHash RepDB Statistics Report" in txt and "  Sketch size:            1000\n" in txt
    assert "  Containment mode:       no\n" in txt and "[Representative Sketch Sizes]" not in txt
    assert "  Total genomes:          %d\n" % len(first) in txt and "  Representatives:        %d\n" % len(reps) in txt
    assert "  Unique hashes:          %d\n" % len(want_index) in txt
    assert "  Total sequence length:  %d bp\n" % (L * len(first)) in txt
    # ---- --query / --assign: every representative sharing a hash, by distance, no filter ----
    rep_sets = [set(sk[first[g]].tolist()) for g in reps]
    qout, aout = os.path.join(tmp, "q.tsv"), os.path.join(tmp, "as.tsv")
    _run([G, "--db", db, "--query", "-l", "-i", lb, "--top-k", "2", "-t", "4", "-o", qout], tmp)
    _run([G, "--db", db, "--assign", "-l", "-i", lb, "-t", "4", "-o", aout], tmp)
    want_q = ["#query\trank\trep_name\tdistance\tcluster_id\tcluster_size\n"]
    want_a = ["#query\tassigned_cluster\trep_name\tdistance\tcluster_size\tstatus\n"]
    n_match = n_none = 0
    for i in second:
        q = set(sk[i].tolist())
        cand = sorted((_mh_dist(len(q & rs), len(q), len(rs), 21), r) for r, rs in enumerate(rep_sets) if q & rs)
        assert len({c[0] for c in cand}) == len(cand)  # no distance ties in the test data
        if not cand:
            want_q.append("%s\t0\tno_match\t-1\t-1\t0\n" % paths[i]); n_none += 1
        for rank, (dist, r) in enumerate(cand[:2]):
            want_q.append("%s\t%d\t%s\t%.6f\t%d\t%d\n" % (paths[i], rank + 1, paths[first[reps[r]]], dist, r, len(clusters[r])))
        if cand and cand[0][0] <= 0.05:
            want_a.append("%s\t%d\t%s\t%.6f\t%d\tassigned\n" % (paths[i], cand[0][1], paths[first[reps[cand[0][1]]]], cand[0][0], len(clusters[cand[0][1]])))
            n_match += 1
        else:
            want_a.append("%s\t-1\tunassigned\t-1\t0\tnovel\n" % paths[i])
    assert n_match >= 2 and n_match + n_none < len(second) and n_none >= 3
    assert open(qout).readlines() == want_q
    assert open(aout).readlines() == want_a
    # ---- --append: MinHashIncrementalCluster (minimum-common filter, no size-ratio filter) ----
    apout = os.path.join(tmp, "ap.out")
    err = _run([G, "--db", db, "--append", lb, "-l", "-t", "4", "-o", apout], tmp)
    assert "===== MinHash RepDB Append =====" in err
    order = first + second
    sets = [set(sk[i].tolist()) for i in order]
    x = math.exp(-0.05 * 21); jmin = x / (2.0 - x)
    for q in range(len(first), len(order)):
        rl = sorted(cid, key=lambda g: cid[g])
        cand = []
        for r, g in enumerate(rl):
            cm = len(sets[q] & sets[g])
            if cm == 0 or cm < int(jmin * (len(sets[q]) + len(sets[g])) / (1.0 + jmin)):
                continue
            dist = _mh_dist(cm, len(sets[q]), len(sets[g]), 21)
            if dist <= 0.05:
                cand.append((dist, r))
        if cand:
            clusters[min(cand)[1]].append(q)
        else:
            cid[q] = len(clusters); clusters.append([])
    assert _parse_clusters(apout) == clusters
    assert open(apout).read().startswith("# Clustering threshold: 0.050000\n# Total clusters: %d\n#\n" % len(clusters))
    d = _read_mh_repdb(db)
    reps = sorted(cid, key=lambda g: cid[g])
    assert any(g >= len(first) for g in reps)
    assert d["rep_ids"] == reps and d["clusters"] == clusters
    assert d["names"] == [paths[i] for i in order] and d["rep_names"] == [paths[order[g]] for g in reps]
    assert all(np.array_equal(a, sk[order[g]]) for a, g in zip(d["rep_sk"], reps))


def test_clust_mst_batching_gzip_retry_and_min_length_filter(oracle, tmp_path):
    """Several small staging batches, a multi-member gzip input whose ISIZE trailer under-reports
    its content (re-parsed in the retry round), and a too-short genome in the middle of the list
    (dropped by -m): sketches must still come out in list order and equal the oracle's."""
    import gzip
    tmp = str(tmp_path)
    L = 2_000_000  # tune_parameters keeps -k 21 at this size
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 2, 3, L, seed=9)
    # genome 1 -> two gzip members; genome 4 -> plain gzip
    raw1 = open(paths[1], "rb").read()
    half = raw1.index(b"\n", len(raw1) // 2) + 1
    with open(paths[1] + ".gz", "wb") as f:
        f.write(gzip.compress(raw1[:half], mtime=0) + gzip.compress(raw1[half:], mtime=0))
    with gzip.GzipFile(paths[4] + ".gz", "wb", mtime=0) as f:
        f.write(open(paths[4], "rb").read())
    short = os.path.join(tmp, "short.fna")
    open(short, "wb").write(b">tiny genome\n" + seqs[0].tobytes()[:5000] + b"\n")
    files = [paths[0], paths[1] + ".gz", short, paths[2], paths[3], paths[4] + ".gz", paths[5]]
    open(lst, "w").write("\n".join(files) + "\n")
    out = os.path.join(tmp, "mst.out")
    env = dict(os.environ, RTC_BATCH_BYTES=str(5_000_000), RTC_VERBOSE="1")
    r = subprocess.run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "1000", "-d", "0.05", "-t", "3",
                        "-o", out], cwd=tmp, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stderr.count("[parse] batch") >= 4  # 3 planned batches + the retry batch
    folder = [os.path.join(tmp, d) for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d)) and d[:2] == "20"][0]
    _, got_sk = _read_hash_sketch(folder)
    keep = [0, 1, 2, 3, 4, 5]
    off = np.arange(len(keep) + 1, dtype=np.uint64) * np.uint64(L)
    want_sk = oracle.sketch_minhash_batch(np.concatenate([seqs[g] for g in keep]), off, 21, 1000)
    assert len(got_sk) == 6
    assert [g for g in range(6) if not np.array_equal(got_sk[g], want_sk[g])] == [], r.stderr[-2000:]
    text = open(out).read()
    assert "short.fna" not in text and paths[1] + ".gz" in text
    flat, start, lens = oracle.to_csr(want_sk)
    want_cl = oracle.forest_clusters(oracle.mst(flat, start, lens, 21, 0, 0.05), 0.05, 6)
    assert _partition(_parse_clusters(out)) == _partition(want_cl)


def test_cli_edge_cases(oracle, tmp_path):
    """Degenerate inputs the command lines must take without crashing: one genome, identical
    genomes (distance 0), blank lines in the list, every genome under -m, a missing file."""
    tmp = str(tmp_path)
    L = 2_000_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 1, 2, L, seed=21)
    exe = os.path.join(BIN, "clust-mst")
    base = ["-k", "21", "-s", "1000", "-d", "0.05", "-t", "2", "-e"]

    def run(list_lines, extra=(), expect_ok=True):
        open(lst, "w").write("\n".join(list_lines) + "\n")
        out = os.path.join(tmp, "o.cluster")
        if os.path.exists(out):
            os.remove(out)
        r = subprocess.run([exe, "-l", "-i", lst, "-o", out] + base + list(extra), cwd=tmp, capture_output=True, text=True, timeout=600)
        assert (r.returncode == 0) == expect_ok, r.stderr[-2000:]
        return (_parse_clusters(out) if expect_ok else None), r.stderr

    # one genome -> one singleton cluster
    cl, _ = run([paths[0]])
    assert cl == [[0]]
    # the same file twice (+ blank lines): distance 0, one cluster of two
    cl, _ = run([paths[0], "", paths[0], ""])
    assert _partition(cl) == [(0, 1)]
    # greedy on the same input
    out = os.path.join(tmp, "g.cluster")
    r = subprocess.run([os.path.join(BIN, "clust-greedy"), "-l", "-i", lst, "-o", out, "-k", "21", "-d", "0.05", "-t", "2", "-e"],
                       cwd=tmp, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _partition(_parse_clusters(out)) == [(0, 1)]
    # everything below the minimum length: a clean error, no crash
    _, err = run([paths[0], paths[1]], extra=["-m", "3000000"], expect_ok=False)
    assert "ERROR" in err or "no genome" in err
    # a missing file: the reference's message and exit(1)
    _, err = run([paths[0], os.path.join(tmp, "does_not_exist.fna")], expect_ok=False)
    assert "cannot open the genome file" in err


def test_clust_mst_config1_kmer_override_64_genomes(oracle, tmp_path):
    """BASELINE config 1: 64 x 1 Mbp, `-k 21 -s 1000`.  tune_parameters replaces 21 by the recommended
    17 (21 > 17 + 3, src/sub_command.cpp:2414-2430): the hash.sketch header must say k = 17 and sketches,
    MST weights and clusters must equal the oracle's at k = 17."""
    tmp = str(tmp_path)
    L = 1_000_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 8, 8, L, seed=8)
    out = os.path.join(tmp, "c1.out")
    err = _run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "1000", "-d", "0.05", "-t", "4", "-o", out], tmp)
    assert "17" in err  # the override is announced
    folder = [os.path.join(tmp, d) for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d)) and d[:2] == "20"][0]
    hdr, got_sk = _read_hash_sketch(folder)
    assert hdr == (0, 17, False, 1000)
    off = np.arange(len(seqs) + 1, dtype=np.uint64) * L
    want_sk = oracle.sketch_minhash_batch(np.concatenate(seqs), off, 17, 1000)
    assert len(got_sk) == 64 and all(np.array_equal(a, b) for a, b in zip(got_sk, want_sk))
    flat, start, lens = oracle.to_csr(want_sk)
    want_mst = oracle.mst(flat, start, lens, 17, 0, 0.05)
    got_mst = _read_edges(folder)
    assert np.array_equal(np.sort(got_mst["dist"]).view(np.uint64), np.sort(want_mst["dist"]).view(np.uint64))
    assert _partition(_parse_clusters(out)) == _partition(oracle.forest_clusters(want_mst, 0.05, 64))
    # the tuner itself agrees with the oracle's restatement on this input
    size = os.path.getsize(paths[0])
    t = oracle.tune_parameters(0, 1, 0, 1, 21, 0.05, 1000, 1000, size, size, size)
    assert t.ok and t.kmer_size == 17


def _check_index_files(folder, fast):
    """The inverted-index files a run leaves behind (minhash.sketch.index: MHIDX001, src/SketchInfo.h:115-160;
    kssd.sketch.index + .dict, src/SketchInfo.cpp:1379-1467) must invert exactly the sketches stored
    beside them: rebuild the posting lists from the files and compare with the sketch file."""
    if not fast:
        _, sk = _read_hash_sketch(folder)
        raw = open(os.path.join(folder, "minhash.sketch.index"), "rb").read()
        assert raw[:8] == b"MHIDX001"
        (H,) = struct.unpack_from("<Q", raw, 8)
        pos, inv = 16, {}
        for _ in range(H):
            h, m = struct.unpack_from("<QI", raw, pos); pos += 12
            inv[h] = sorted(struct.unpack_from(f"<{m}I", raw, pos)); pos += 4 * m
        assert pos == len(raw)
    else:
        raw = open(os.path.join(folder, "kssd.hash.sketch"), "rb").read()
        n = struct.unpack_from("<iiiii", raw, 0)[4]
        pos, sk = 20, []
        for _ in range(n):
            (m,) = struct.unpack_from("<Q", raw, pos); pos += 8
            sk.append(np.frombuffer(raw, dtype="<u4", count=m, offset=pos)); pos += 4 * m
        idx = open(os.path.join(folder, "kssd.sketch.index"), "rb").read()
        (H,) = struct.unpack_from("<Q", idx, 0)
        keys = np.frombuffer(idx, dtype="<u4", count=H, offset=8)
        counts = np.frombuffer(idx, dtype="<u4", count=H, offset=8 + 4 * H)
        ids = np.frombuffer(open(os.path.join(folder, "kssd.sketch.dict"), "rb").read(), dtype="<u4")
        assert len(idx) == 8 + 8 * H and counts.sum() == len(ids)
        inv, p = {}, 0
        for k_, c in zip(keys.tolist(), counts.tolist()):
            inv[k_] = sorted(ids[p:p + c].tolist()); p += c
    want = {}
    for gi, h in enumerate(sk):
        for x in h.tolist():
            want.setdefault(x, []).append(gi)
    assert inv == want and len(inv) > 0


@pytest.mark.parametrize("fast", [False, True])
def test_clust_mst_append_equals_full_run(oracle, tmp_path, fast):
    """--append (append_clust_mst, src/sub_command.cpp:1532-1759): cluster 9 genomes, append 7 more to the
    stored folder, compare with one run over all 16 -- same sketch file bytes, same MST weights
    (bit-for-bit, also equal to the oracle's), same clusters; the appended folder can be resumed from."""
    tmp = str(tmp_path)
    L = 1_800_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 4, 4, L, seed=21)
    perm = [0, 5, 10, 15, 1, 4, 8, 12, 13, 2, 3, 6, 7, 9, 11, 14]  # families straddle the split
    first, second = [paths[i] for i in perm[:9]], [paths[i] for i in perm[9:]]
    la, lb, lall = (os.path.join(tmp, x) for x in ("a.txt", "b.txt", "all.txt"))
    open(la, "w").write("\n".join(first) + "\n")
    open(lb, "w").write("\n".join(second) + "\n")
    open(lall, "w").write("\n".join(first + second) + "\n")
    mode = ["--fast"] if fast else ["-s", "800"]
    skname = "kssd.hash.sketch" if fast else "hash.sketch"

    def run(sub, args):
        d = os.path.join(tmp, sub)
        os.makedirs(d)
        out = os.path.join(d, "res.out")
        err = _run([os.path.join(BIN, "clust-mst")] + args + ["-d", "0.05", "-t", "4", "-o", out], d)
        folders = sorted(os.path.join(d, x) for x in os.listdir(d) if os.path.isdir(os.path.join(d, x)))
        return out, folders, err

    out_all, f_all, _ = run("all", ["-l", "-i", lall, "-k", "21"] + mode)
    out_a, f_a, _ = run("a", ["-l", "-i", la, "-k", "21"] + mode)
    out_ab, f_ab, err = run("ab", ["-l", "--append", lb, "--presketched", f_a[0]] + mode[:1] * fast)
    assert "---the start_index is: 9" in err
    for folder in (f_all[0], f_a[0], f_ab[0]):
        _check_index_files(folder, fast)
    assert open(os.path.join(f_ab[0], skname), "rb").read() == open(os.path.join(f_all[0], skname), "rb").read()
    got, want = _read_edges(f_ab[0]), _read_edges(f_all[0])
    assert len(got) == len(want) and np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
    assert _partition(_parse_clusters(out_ab)) == _partition(_parse_clusters(out_all))
    assert len(_parse_clusters(out_ab)) < 16
    if not fast:  # the oracle's MST over all 16 sketches has the same weights
        _, sk = _read_hash_sketch(f_ab[0])
        flat, start, lens = oracle.to_csr(sk)
        omst = oracle.mst(flat, start, lens, 21, 0, 0.05)
        assert np.array_equal(np.sort(got["dist"]).view(np.uint64), np.sort(omst["dist"]).view(np.uint64))
    # the combined folder resumes like any other
    out_r = os.path.join(tmp, "resume.out")
    _run([os.path.join(BIN, "clust-mst"), "--premsted", f_ab[0], "-d", "0.05", "-o", out_r] + mode[:1] * fast, tmp)
    assert open(out_r).read() == open(out_ab).read()
    # usage errors of the reference (src/main.cpp:643-646)
    r = subprocess.run([os.path.join(BIN, "clust-mst"), "-l", "--append", lb, "-o", out_r], capture_output=True, text=True)
    assert r.returncode != 0 and "--presketched or --premsted needed" in r.stderr


def test_clust_mst_dense_files_and_noise_removal(oracle, tmp_path):
    """--dense (src/sub_command.cpp:3071-3103): mst.dense / mst.ani hold the brute-force histograms of the
    candidate pairs, the .removeNoise file is the forest without the edges of low-density nodes; --premsted
    --dense reproduces it from the stored files."""
    from test_gpu_mst import _dense_brute_force
    tmp = str(tmp_path)
    L = 1_800_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 6, L, seed=31)
    out = os.path.join(tmp, "d.out")
    _run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "500", "-d", "0.05", "--dense", "-t", "4", "-o", out], tmp)
    folder = [os.path.join(tmp, d) for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d)) and d[:2] == "20"][0]
    _, sk = _read_hash_sketch(folder)
    n = len(sk)
    want_dense, want_ani = _dense_brute_force(oracle, sk, 21, False, 0.05)
    raw = open(os.path.join(folder, "mst.dense"), "rb").read()
    gn, span = struct.unpack_from("<ii", raw, 0)
    assert (gn, span) == (n, 100) and len(raw) == 8 + 4 * n * 100
    assert np.array_equal(np.frombuffer(raw, dtype="<i4", offset=8).reshape(100, n), want_dense)
    assert np.array_equal(np.frombuffer(open(os.path.join(folder, "mst.ani"), "rb").read(), dtype="<u8"), want_ani)
    # the noise pass restated: per multi-member cluster, nodes with density <= max(min(Q1 - 1, 2), 0) at bucket d / 0.01
    mst = _read_edges(folder)
    forest = [e for e in mst if e["dist"] <= 0.05]

    def components(edges):
        parent = list(range(n))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        for e in edges:
            parent[find(int(e["pre"]))] = find(int(e["suf"]))
        groups = {}
        for v in range(n):
            groups.setdefault(find(v), []).append(v)
        return list(groups.values())
    noise = set()
    idx = int(0.05 / 0.01)
    for cl in components(forest):
        if len(cl) == 1:
            continue
        dens = sorted(int(want_dense[idx, v]) for v in cl)
        thr = max(min(dens[len(dens) // 4] - 1, 2), 0)
        noise |= {v for v in cl if want_dense[idx, v] <= thr}
    kept = [e for e in forest if int(e["pre"]) not in noise and int(e["suf"]) not in noise]
    assert _partition(_parse_clusters(out + ".removeNoise")) == _partition(components(kept))
    assert not open(out + ".removeNoise").read().startswith("#")  # printed without the threshold header
    out2 = os.path.join(tmp, "d2.out")
    _run([os.path.join(BIN, "clust-mst"), "--premsted", folder, "--dense", "-d", "0.05", "-o", out2], tmp)
    assert open(out2 + ".removeNoise").read() == open(out + ".removeNoise").read()


def test_single_fasta_sequence_mode(oracle, tmp_path):
    """Without -l the input is one FASTA file and every record of at least -m bases is a genome (sketchSequences /
    sketchSequencesWithKssd, src/SketchInfo.cpp:554-862; calSize's sequence branch :484-535): records below the
    minimum are dropped, the rest keep file order; info.sketch in its sequence layout (src/Sketch_IO.cpp:88-134),
    sketches equal to the oracle's, clusters printed in the sequence format (src/MST_IO.cpp:136-160).  clust-mst,
    clust-greedy, MinHash and --fast, plus the --presketched flow of the folder."""
    from rabbittclust_amd import api
    tmp = str(tmp_path)
    L = 2_000_000
    desc = api.synth_family_descs(3, 3, global_seed=77)
    seqs = [oracle.synth_genome(int(d["fam_seed"]), int(d["mut_seed"]), int(d["mut_thr"]), L) for d in desc]
    lens = [L, L - 1234, L - 77, L - 50000, L, L - 9, L - 300000, L - 1, L - 4321]   # distinct but for two: ties keep file order
    fa = os.path.join(tmp, "all.fna")
    recs = []
    with open(fa, "wb") as f:
        for g, s in enumerate(seqs):
            body = s.tobytes()[:lens[g]]
            if g == 4:
                body = body.lower()
            name, comment = "seq%d" % g, ("family %d member" % (g // 3) if g % 2 == 0 else None)
            f.write((">" + name + (" " + comment if comment else "") + "\n").encode())
            w = 60 if g % 2 else 100
            for i in range(0, len(body), w):
                f.write(body[i:i + w] + b"\n")
            recs.append((name, comment or "", body))
            if g in (1, 5):   # short records between the genomes: dropped by -m
                f.write((">short%d tiny\n" % g).encode() + body[:5000] + b"\n")
    M = os.path.join(BIN, "clust-mst")
    out = os.path.join(tmp, "mst.out")
    dm = os.path.join(tmp, "m"); os.makedirs(dm)
    err = _run([M, "-i", fa, "-k", "21", "-s", "1000", "-d", "0.05", "-m", "10000", "-t", "4", "-o", out], dm)
    assert "\t===the genome number for clustering is: 9" in err and "threshold is: 2\n" in err
    assert "\t===the maxSize is: %d" % L in err and "\t===the minSize is: %d" % (L - 300000) in err
    folder = [os.path.join(dm, d) for d in os.listdir(dm) if os.path.isdir(os.path.join(dm, d))][0]
    # info.sketch, sequence layout
    raw = open(os.path.join(folder, "info.sketch"), "rb").read()
    by_file, n = struct.unpack_from("<?Q", raw, 0)
    assert (by_file, n) == (False, 9)
    p = 9
    for name, comment, body in recs:
        nl, cl, strand, length = struct.unpack_from("<iiii", raw, p); p += 16
        assert (nl, cl, strand, length) == (len(name), len(comment), 0, len(body))
        assert raw[p:p + nl].decode() == name and raw[p + nl:p + nl + cl].decode() == comment
        p += nl + cl
    assert p == len(raw)
    off = np.zeros(10, dtype=np.uint64); off[1:] = np.cumsum([len(r[2]) for r in recs])
    allb = np.frombuffer(b"".join(r[2] for r in recs), dtype=np.uint8)
    want_sk = oracle.sketch_minhash_batch(allb, off, 21, 1000)
    hdr, got_sk = _read_hash_sketch(folder)
    assert hdr == (0, 21, False, 1000)
    assert len(got_sk) == 9 and all(np.array_equal(a, b) for a, b in zip(got_sk, want_sk))
    flat, start, slens = oracle.to_csr(want_sk)
    want_mst = oracle.mst(flat, start, slens, 21, 0, 0.05)
    got_cl = _parse_seq_clusters(out)
    assert _partition(got_cl) == _partition(oracle.forest_clusters(want_mst, 0.05, 9)) and 1 < len(got_cl) < 9
    text = open(out).read()
    assert ("\t%6d\t%6d\t%12dnt\t%20s\t%s\n" % (0, 0, lens[0], "seq0", "family 0 member")) in text
    assert ("\t%6d\t%12dnt\t%20s\t%s\n" % (1, lens[1], "seq1", "")) in text   # no comment: an empty field
    out2 = os.path.join(tmp, "mst2.out")
    _run([M, "--presketched", folder, "-d", "0.05", "-o", out2], tmp)
    assert open(out2).read() == text
    # clust-greedy on the same file: greedy pass in file order
    G = os.path.join(BIN, "clust-greedy")
    og = os.path.join(tmp, "g.out")
    dg = os.path.join(tmp, "g"); os.makedirs(dg)
    _run([G, "-i", fa, "-k", "21", "-s", "1000", "-d", "0.05", "-t", "4", "-o", og], dg)
    ncl, rep = oracle.greedy_minhash(flat, start, slens, 1000, 21, False, 0.05)
    want = {}
    for i, r in enumerate(rep):
        want.setdefault(int(r), []).append(i)
    assert _parse_seq_clusters(og) == [[r] + [m for m in ms if m != r] for r, ms in sorted(want.items())]
    # clust-greedy's default: containment sketches, size = record length / (average length / 1000) (tune_parameters)
    oc = os.path.join(tmp, "gc.out")
    _run([G, "-i", fa, "-k", "21", "-d", "0.05", "-t", "4", "-e", "-o", oc], tmp)
    compress = (sum(lens) // len(lens)) // 1000
    cfg = np.array([max(x // compress, 100) for x in lens], dtype=np.uint32)
    csk = oracle.sketch_minhash_batch(allb, off, 21, cfg)
    cflat, cstart, clens = oracle.to_csr(csk)
    ncl, rep = oracle.greedy_minhash(cflat, cstart, clens, cfg, 21, True, 0.05)
    want = {}
    for i, r in enumerate(rep):
        want.setdefault(int(r), []).append(i)
    assert _parse_seq_clusters(oc) == [[r] + [m for m in ms if m != r] for r, ms in sorted(want.items())]
    # --fast: KSSD sketches per record
    of = os.path.join(tmp, "f.out")
    df = os.path.join(tmp, "f"); os.makedirs(df)
    _run([M, "--fast", "-i", fa, "-k", "21", "-d", "0.05", "-t", "4", "-o", of], df)
    ffolder = [os.path.join(df, d) for d in os.listdir(df) if os.path.isdir(os.path.join(df, d))][0]
    want_ks = [oracle.kssd_sketch(np.frombuffer(r[2], dtype=np.uint8), 21, 3) for r in recs]
    rawk = open(os.path.join(ffolder, "kssd.hash.sketch"), "rb").read()
    assert struct.unpack_from("<iiiii", rawk, 0) == ((11 << 8) + (6 << 4) + 3, 11, 6, 3, 9)
    pos = 20
    for w in want_ks:
        (m,) = struct.unpack_from("<Q", rawk, pos); pos += 8
        assert np.array_equal(np.frombuffer(rawk, dtype="<u4", count=m, offset=pos), w); pos += 4 * m
    assert pos == len(rawk)
    kflat, kstart, klens = oracle.to_csr(want_ks, dtype=np.uint32)
    assert _partition(_parse_seq_clusters(of)) == _partition(oracle.forest_clusters(oracle.mst(kflat, kstart, klens, 22, 0, 0.05), 0.05, 9))
    # a list file given without -l is refused like any non-FASTA name
    r = subprocess.run([M, "-i", os.path.join(tmp, "list.txt"), "-o", out2], capture_output=True, text=True)
    assert r.returncode != 0 and "Only support FASTA files" in r.stderr


def _parse_seq_clusters(path):
    clusters, cur = [], None
    for ln in open(path):
        if ln.startswith("the cluster"):
            cur = []
            clusters.append(cur)
        elif ln.startswith("\t"):
            cur.append(int(ln.split("\t")[2]))
    return clusters


def test_packed_staging_unpack_kernel_and_cli_identity(ctx, oracle, tmp_path):
    """(1) rtc_unpack_bases_dev against its restatement on a packed batch with runs at every alignment.
    (2) The command lines stage 2-bit packed bases by default and sketch them as they are (rtc_sketch_minhash_packed_dev,
    rtc_sketch_kssd_packed_dev): hash.sketch / kssd.hash.sketch / edge.mst / the cluster text must be byte-identical to
    the RTC_STAGE_ASCII=1 run and to the RTC_SKETCH_UNPACK=1 / RTC_KSSD_UNPACK=1 run on genomes with N runs, IUPAC codes, lower case,
    several records, CRLF line ends and a gzip member, over several small batches."""
    import ctypes as C
    import gzip
    import torch
    from rabbittclust_amd import _lib
    rng = np.random.default_rng(11)
    n = 64 * 2000
    packed = rng.integers(0, 256, size=n // 4, dtype=np.uint8)
    starts = np.sort(rng.choice(n - 200, size=400, replace=False))
    runs = []
    for st_ in starts:
        ln = int(rng.integers(1, 130))
        if runs and runs[-2] + runs[-1] >= st_:
            continue
        runs += [int(st_), ln]
    runs += [n - 64, 64]
    d_p = torch.from_numpy(packed).to(ctx.device)
    d_r = torch.from_numpy(np.array(runs, dtype=np.uint64).view(np.int64)).to(ctx.device)
    d_o = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
    ctx.check(ctx.lib.rtc_unpack_bases_dev(ctx.h, C.c_void_p(d_p.data_ptr()), n, C.c_void_p(d_r.data_ptr()), len(runs) // 2, C.c_void_p(d_o.data_ptr())))
    ctx.sync()
    codes = np.stack([(packed >> (2 * b)) & 3 for b in range(4)], axis=1).reshape(-1)
    want = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
    for st_, ln in zip(runs[0::2], runs[1::2]):
        want[st_:st_ + ln] = ord("N")
    assert np.array_equal(d_o.cpu().numpy(), want)
    assert ctx.lib.rtc_unpack_bases_dev(ctx.h, C.c_void_p(d_p.data_ptr()), 100, None, 0, C.c_void_p(d_o.data_ptr())) == _lib.RTC_ERR_ARG
    # ---- command lines ----
    tmp = str(tmp_path)
    L = 1_900_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 4, L, seed=91, two_records=True)
    for g, p in enumerate(paths):   # damage the files: N runs, IUPAC codes, lower case, CRLF, one gzip
        raw = bytearray(open(p, "rb").read())
        body = [i for i, c in enumerate(raw) if c in b"ACGT"]
        sel = rng.choice(len(body), size=40, replace=False)
        for j in sel[:25]:
            a = body[j]
            for q in range(a, min(a + int(rng.integers(1, 90)), len(raw))):
                if raw[q] in b"ACGT":
                    raw[q] = ord("N")
        for j in sel[25:]:
            raw[body[j]] = int(rng.choice(np.frombuffer(b"RYKMSWn", dtype=np.uint8)))
        low = rng.random(len(raw)) < (0.3 if g % 2 else 0.0)
        for q in np.nonzero(low)[0]:
            if raw[q] in b"ACGT":
                raw[q] |= 0x20
        data = bytes(raw).replace(b"\n", b"\r\n") if g == 1 else bytes(raw)
        if g == 2:
            p2 = p + ".gz"
            with gzip.GzipFile(p2, "wb", mtime=0) as f:
                f.write(data)
            paths[g] = p2
        else:
            open(p, "wb").write(data)
    open(lst, "w").write("\n".join(paths) + "\n")
    for tool, extra in (("clust-mst", ["-s", "500"]), ("clust-mst", ["--fast"]), ("clust-greedy", ["-c", "1000"])):
        outs = {}
        # the sketch kernels read the packed stream themselves; RTC_SKETCH_UNPACK=1 (--fast: also the older RTC_KSSD_UNPACK=1)
        # expands every batch to characters in HBM first
        variants = [("packed", {"RTC_VERBOSE": "1"}), ("ascii", {"RTC_STAGE_ASCII": "1", "RTC_VERBOSE": "1"}),
                    ("unpack", {"RTC_KSSD_UNPACK" if extra == ["--fast"] else "RTC_SKETCH_UNPACK": "1", "RTC_VERBOSE": "1"})]
        for tag, env in variants:
            d = os.path.join(tmp, tool + extra[0].strip("-") + tag)
            os.makedirs(d)
            out = os.path.join(d, "res.out")
            r = subprocess.run([os.path.join(BIN, tool), "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "-o", out] + extra, cwd=d,
                               capture_output=True, text=True, timeout=600, env=dict(os.environ, RTC_BATCH_BYTES=str(5 << 20), **env))
            assert r.returncode == 0, r.stderr[-2000:]
            folder = [os.path.join(d, x) for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))][0]
            files = sorted(f for f in os.listdir(folder))
            outs[tag] = (open(out).read(), {f: open(os.path.join(folder, f), "rb").read() for f in files})
            assert ("over packed bases" in r.stderr) == (tag == "packed"), r.stderr[-2000:]
        assert outs["unpack"] == outs["packed"]
        assert outs["packed"][0] == outs["ascii"][0], (tool, extra)
        assert outs["packed"][1].keys() == outs["ascii"][1].keys()
        for f in outs["packed"][1]:
            assert outs["packed"][1][f] == outs["ascii"][1][f], (tool, extra, f)


@pytest.mark.parametrize("mode", [["-s", "500"], ["--fast"]])
def test_batches_parsed_while_the_gpus_come_up_change_nothing(oracle, tmp_path, mode):
    """(--fast: the packed batches are sketched as they are, retry round included.)  The command line parses batch after batch into buffers of their own while the HIP runtime and the contexts come up on
    another thread, and hands them to the lanes afterwards.  With small batches (many of them before the GPUs are there),
    with the pre-parse switched off (RTC_PREPARSE_BYTES=0) and with a budget that stops it after a few batches, the
    cluster file, hash.sketch and edge.mst must be byte-identical -- including a gzip file whose slot guess is too small
    (retry round) and a genome below the length filter."""
    import gzip
    import re
    tmp = str(tmp_path)
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 5, 4, 300_000, seed=91)
    short = os.path.join(tmp, "short.fna")
    open(short, "wb").write(b">short x\n" + b"ACGT" * 1000 + b"\n")
    gz = os.path.join(tmp, "g_extra.fna.gz")
    with gzip.open(gz, "wb") as f:   # two members: the trailing ISIZE covers only the last one
        f.write(open(paths[3], "rb").read())
    with open(gz, "ab") as f:
        f.write(gzip.compress(b">tail y\n" + b"ACGTTGCA" * 20000 + b"\n"))
    open(lst, "a").write(short + "\n" + gz + "\n")
    outs = []
    for name, env in (("pre", {"RTC_BATCH_BYTES": "2000000"}), ("off", {"RTC_BATCH_BYTES": "2000000", "RTC_PREPARSE_BYTES": "0"}),
                      ("few", {"RTC_BATCH_BYTES": "2000000", "RTC_PREPARSE_BYTES": "1500000"}),
                      ("many", {"RTC_BATCH_BYTES": "2000000", "RTC_PREPARSE_BYTES": "100000000"})):  # beyond the three ring buffers
        d = os.path.join(tmp, name)
        os.makedirs(d)
        err = _run([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "-o", os.path.join(d, "out.cluster")] + mode,
                   d, dict(env, RTC_VERBOSE="1"))
        m = re.search(r"\[init\]\s+(\d+) of (\d+) batches parsed before the GPUs were up", err)
        assert m, err[-2000:]
        npre, nb = int(m.group(1)), int(m.group(2))
        assert nb >= 3
        if name == "off":
            assert npre == 0
        if name in ("few", "pre"):   # the default budget is the three buffers that become the staging ring
            assert npre <= 3
        folder = [x for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))]
        assert len(folder) == 1
        fd = os.path.join(d, folder[0])
        outs.append((open(os.path.join(d, "out.cluster"), "rb").read(), open(os.path.join(fd, "kssd.hash.sketch" if mode == ["--fast"] else "hash.sketch"), "rb").read(),
                     open(os.path.join(fd, "edge.mst"), "rb").read(), npre))
    assert outs[0][:3] == outs[1][:3] == outs[2][:3] == outs[3][:3]
    assert outs[0][3] >= 1  # the default budget parsed at least the first batch ahead of the GPUs
