"""CPU suite: host side of the drop-in -- FASTA reader vs the reference's kseq (golden dumps made
with oracle/_ref, plus a live comparison when the harness is present), on-disk formats written
independently here per SURVEY Appendix A and re-saved by the host library byte-for-byte,
--premsted cluster text, parameter tuning vs the oracle, KSSD shuffle table vs the oracle."""
import ctypes as C
import glob
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HOSTLIB = os.path.join(ROOT, "rabbittclust_amd", "librtclust_host.so")


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(HOSTLIB):
        pytest.fail("librtclust_host.so missing: run __graft_entry__.build()")
    lib = C.CDLL(HOSTLIB)
    lib.rtch_fasta_dump.restype = C.c_long
    lib.rtch_premsted.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, C.c_int]
    lib.rtch_tune.argtypes = [C.c_int] * 5 + [C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return lib


def _dump(lib, fn, path):
    need = fn(path.encode(), None, 0)
    if need < 0:
        return None
    buf = C.create_string_buffer(max(need, 1))
    fn(path.encode(), buf, need)
    return buf.raw[:need]


def test_fasta_reader_matches_reference_kseq_golden(host):
    files = sorted(glob.glob(os.path.join(GOLD, "fasta", "*")))
    assert len(files) >= 7
    for f in files:
        want = open(os.path.join(GOLD, "kseq_dump_" + os.path.basename(f) + ".txt"), "rb").read()
        assert _dump(host, host.rtch_fasta_dump, f) == want, f
    assert host.rtch_fasta_dump(b"/nonexistent/file.fa", None, 0) == -1


def test_fasta_reader_matches_reference_kseq_live(host, tmp_path):
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = C.CDLL(ref_path)
    ref.ref_kseq_dump.restype = C.c_long
    rng = np.random.default_rng(9)
    for t in range(20):
        parts = []
        for r in range(int(rng.integers(0, 5))):
            hdr = ">" + "".join(rng.choice(list("abcXYZ_ \t|.:"), size=int(rng.integers(0, 20))))
            eol = "\r\n" if rng.random() < 0.3 else "\n"
            parts.append(hdr + eol)
            for _ in range(int(rng.integers(0, 4))):
                parts.append("".join(rng.choice(list("ACGTNacgt"), size=int(rng.integers(0, 90)))) + eol)
                if rng.random() < 0.2:
                    parts.append(eol)
        p = tmp_path / f"r{t}.fa"
        p.write_bytes("".join(parts).encode())
        assert _dump(host, host.rtch_fasta_dump, str(p)) == _dump(ref, ref.ref_kseq_dump, str(p)), "".join(parts)
    # gzip: our reader inflates a whole file with libdeflate where the host has it (zlib otherwise, and for everything
    # libdeflate does not take); the reference's kseq reads through zlib's gzread.  One member, two members, many small
    # members (bgzip style), a file cut short, bytes behind the last member, an empty member: the same records either way.
    import gzip
    body = "".join(">rec%d some text\n" % r + "\n".join("".join(rng.choice(list("ACGTNacgt"), size=70)) for _ in range(400)) + "\n"
                   for r in range(5)).encode()
    one = gzip.compress(body, 6)
    cases = {
        "one.fa.gz": one,
        "two.fa.gz": gzip.compress(body[:70_000], 6) + gzip.compress(body[70_000:], 1),
        "many.fa.gz": b"".join(gzip.compress(body[i:i + 9_000], 4) for i in range(0, len(body), 9_000)),
        "cut.fa.gz": one[:-1500],
        "trail.fa.gz": one + b"trailing bytes that are no gzip member",
        "hole.fa.gz": gzip.compress(body[:50_000], 6) + gzip.compress(b"", 6) + gzip.compress(body[50_000:], 6),
    }
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        got, want = _dump(host, host.rtch_fasta_dump, str(p)), _dump(ref, ref.ref_kseq_dump, str(p))
        assert got == want, name
        assert name in ("cut.fa.gz",) or (want is not None and len(want) > len(body) // 2), name


# ---- independent writers of the on-disk formats ----
def _write_info(path, genomes, kssd=False, use64=False):
    with open(path, "wb") as f:
        f.write(struct.pack("<?Q", True, len(genomes)))
        for fn, name, cm, length in genomes:
            f.write(struct.pack("<iiiiQ", len(fn), len(name), len(cm), 0, length))
            f.write(fn.encode() + name.encode() + cm.encode())
            if kssd:
                f.write(struct.pack("<?", use64))


def _genomes(n):
    return [(f"/data/genome_{i}.fna", f"seq{i}", f"comment number {i}" if i % 3 else "noName", 5_000_000 + 17 * i)
            for i in range(n)]


def test_minhash_folder_roundtrip_bytes(host, tmp_path):
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir(); dst.mkdir()
    rng = np.random.default_rng(11)
    g = _genomes(9)
    sk = [np.unique(rng.integers(0, 1 << 20, size=int(rng.integers(0, 60)), dtype=np.uint64)) for _ in g]
    sk[4] = np.zeros(0, dtype=np.uint64)
    _write_info(src / "info.sketch", g)
    for containment, val in ((False, 1000), (True, 777)):
        with open(src / "hash.sketch", "wb") as f:
            f.write(struct.pack("<ii?i", 0, 21, containment, val))
            for h in sk:
                f.write(struct.pack("<Q", len(h)) + h.astype("<u8").tobytes())
        assert host.rtch_resave_folder(str(src).encode(), str(dst).encode(), 0) == 0
        for name in ("info.sketch", "hash.sketch"):
            assert (src / name).read_bytes() == (dst / name).read_bytes(), name
        # MHIDX001 index: magic, count, then {hash, m, ids[m]} -- check it inverts the sketches
        raw = (dst / "minhash.sketch.index").read_bytes()
        assert raw[:8] == b"MHIDX001"
        (H,) = struct.unpack_from("<Q", raw, 8)
        pos, inv = 16, {}
        for _ in range(H):
            h, m = struct.unpack_from("<QI", raw, pos); pos += 12
            inv[h] = list(struct.unpack_from(f"<{m}I", raw, pos)); pos += 4 * m
        assert pos == len(raw)
        want = {}
        for gi, h in enumerate(sk):
            for x in h.tolist():
                want.setdefault(x, []).append(gi)
        assert inv == want


@pytest.mark.parametrize("use64", [False, True])
def test_kssd_folder_roundtrip_bytes(host, tmp_path, use64):
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir(); dst.mkdir()
    rng = np.random.default_rng(12)
    g = _genomes(7)
    dt = np.uint64 if use64 else np.uint32
    sk = [np.unique(rng.integers(0, 1 << 18, size=int(rng.integers(1, 50)), dtype=np.uint64)).astype(dt) for _ in g]
    _write_info(src / "kssd.info.sketch", g, kssd=True, use64=use64)
    half_k, half_subk, dr = (16, 6, 3) if use64 else (11, 6, 3)
    with open(src / "kssd.hash.sketch", "wb") as f:
        f.write(struct.pack("<iiiii", (half_k << 8) + (half_subk << 4) + dr, half_k, half_subk, dr, len(g)))
        for h in sk:
            f.write(struct.pack("<Q", len(h)) + h.tobytes())
    assert host.rtch_resave_folder(str(src).encode(), str(dst).encode(), 1) == 0
    for name in ("kssd.info.sketch", "kssd.hash.sketch"):
        assert (src / name).read_bytes() == (dst / name).read_bytes(), name
    idx = (dst / "kssd.sketch.index").read_bytes()
    (H,) = struct.unpack_from("<Q", idx, 0)
    w = 8 if use64 else 4
    keys = np.frombuffer(idx, dtype=dt, count=H, offset=8)
    counts = np.frombuffer(idx, dtype=np.uint32, count=H, offset=8 + w * H)
    assert len(idx) == 8 + (w + 4) * H
    ids = np.frombuffer((dst / "kssd.sketch.dict").read_bytes(), dtype=np.uint32)
    assert counts.sum() == len(ids) == sum(len(h) for h in sk)
    pos = 0
    for k_, c in zip(keys.tolist(), counts.tolist()):
        assert sorted(ids[pos:pos + c].tolist()) == [gi for gi, h in enumerate(sk) if k_ in h.tolist()]
        pos += c


def test_premsted_cluster_text_and_mst_roundtrip(host, tmp_path, oracle):
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir(); dst.mkdir()
    g = _genomes(8)
    _write_info(src / "info.mst", g)
    edges = [(1, 0, 0.01), (2, 1, 0.04), (5, 4, 0.0), (6, 2, 0.2), (7, 6, 0.05)]
    with open(src / "edge.mst", "wb") as f:
        f.write(struct.pack("<Q", len(edges)))
        for a, b, d in edges:
            f.write(struct.pack("<iid", a, b, d))
    out = tmp_path / "result.out"
    assert host.rtch_premsted(str(src).encode(), str(dst).encode(), str(out).encode(), 0.05, 0) == 0
    assert (src / "edge.mst").read_bytes() == (dst / "edge.mst").read_bytes()
    assert (src / "info.mst").read_bytes() == (dst / "info.mst").read_bytes()
    text = out.read_text().splitlines()
    assert text[0] == "# Clustering threshold: 0.050000" and text[1] == "# Total clusters: 4" and text[2] == "#"
    # clusters in BFS order from ascending ids: {0,1,2}, {3}, {4,5}, {6,7}
    e = np.array(edges, dtype=oracle.EDGE_DT)
    want = oracle.forest_clusters(e, 0.05, 8)
    assert want == [[0, 1, 2], [3], [4, 5], [6, 7]]
    body = [ln for ln in text[3:] if ln]
    assert body[0] == "the cluster 0 is: "
    fn, name, cm, length = g[0]
    assert body[1] == "\t%5d\t%6d\t%12dnt\t%20s\t%20s\t%s" % (0, 0, length, fn, name, cm)
    ids = [int(ln.split("\t")[2]) for ln in body if ln.startswith("\t")]
    assert ids == [x for c in want for x in c]


def test_tune_parameters_matches_oracle(host, oracle):
    cases = [(0, 1, 0, 1, 21, 0.05, 1000, 1000, 1012520, 1012520, 1012520),
             (0, 1, 0, 1, 21, 0.05, 1000, 1000, 5062520, 4000000, 4800000),
             (1, 0, 0, 0, 19, 0.05, 1000, 1000, 5062520, 3000000, 4100000),
             (1, 1, 1, 0, 21, 0.05, 9000000, 1000, 5062520, 3000000, 4100000),
             (0, 1, 0, 1, 12, 0.05, 1000, 1000, 2025040, 2025040, 2025040),
             (0, 1, 0, 1, 21, 0.6, 1000, 1000, 5062520, 5062520, 5062520),
             (0, 1, 1, 1, 21, 0.05, 1000, 1000, 5062520, 5062520, 5062520)]
    for cs in cases:
        r = oracle.tune_parameters(*cs)
        k, cc, ic = C.c_int(), C.c_int(), C.c_int()
        ok = host.rtch_tune(*cs[:5], cs[5], cs[6], cs[7], cs[8], cs[9], cs[10], C.byref(k), C.byref(cc), C.byref(ic))
        assert ok == r.ok, cs
        if r.ok:
            assert (k.value, cc.value, ic.value) == (r.kmer_size, r.contain_compress, r.is_containment), cs


def test_shuffle_dim_matches_oracle_and_file_sizes(host, oracle, tmp_path):
    out = np.zeros(1 << 24, dtype=np.int32)
    assert host.rtch_shuffle_dim(6, out.ctypes.data_as(C.c_void_p)) == 1 << 24
    assert np.array_equal(out, oracle.kssd_shuffle_dim(6))
    # calSize / containment file length: plain = stat size, gz = ISIZE trailer
    import gzip
    payload = b">x\n" + b"ACGT" * 5000 + b"\n"
    p1, p2 = tmp_path / "a.fna", tmp_path / "b.fna.gz"
    p1.write_bytes(payload)
    with gzip.open(p2, "wb") as f:
        f.write(payload)
    assert host.rtch_file_length(str(p1).encode()) == len(payload)
    assert host.rtch_file_length(str(p2).encode()) == len(payload)
    lst = tmp_path / "list.txt"
    lst.write_text(f"{p1}\n{p2}\n")
    mx, mn, avg = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert host.rtch_cal_size(str(lst).encode(), C.c_uint64(10000), C.byref(mx), C.byref(mn), C.byref(avg)) == 1
    assert mx.value == mn.value == avg.value == len(payload)


def test_flat_genome_reader_matches_string_reader(host, tmp_path):
    """The zero-copy reader the CLI parses into pinned memory with yields the same byte stream as
    the std::string reader (itself pinned on the reference's kseq dumps above), for every golden
    FASTA/FASTQ/gzip file, and reports the capacity it needs when the slot is too small."""
    import glob
    import gzip
    host.rtch_genome_bases.restype = C.c_long
    host.rtch_genome_bases.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]

    def run(path, flat, cap):
        buf = np.zeros(max(cap, 1), dtype=np.uint8)
        tot, nrec, fl, slot = C.c_uint64(), C.c_uint64(), C.c_int(), C.c_uint64()
        n = host.rtch_genome_bases(str(path).encode(), flat, buf.ctypes.data_as(C.c_void_p), cap, C.byref(tot),
                                   C.byref(nrec), C.byref(fl), C.byref(slot))
        return n, bytes(buf[:max(min(n, cap), 0)]), tot.value, nrec.value, fl.value, slot.value

    files = sorted(glob.glob(os.path.join(GOLD, "fasta", "*")))
    assert len(files) >= 6
    rng = np.random.default_rng(3)
    big = tmp_path / "big.fa"
    seq = rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), size=700_000).tobytes()
    with open(big, "wb") as f:  # lines longer than the reader's buffer, CRLF, blank lines, no final EOL
        f.write(b">r1 first\r\n" + seq[:300_000] + b"\r\n\n>r2\n" + seq[300_000:300_070] + b"\n" + seq[300_070:])
    biggz = tmp_path / "big.fa.gz"
    with gzip.GzipFile(biggz, "wb", mtime=0) as f:
        f.write(open(big, "rb").read())
    multi = tmp_path / "multi.fa.gz"  # two gzip members: the ISIZE trailer under-reports the content
    with open(multi, "wb") as f:
        f.write(gzip.compress(b">a x\n" + seq[:5000] + b"\n", mtime=0) + gzip.compress(b">b\n" + seq[5000:5100] + b"\n", mtime=0))
    for path in files + [big, biggz, multi]:
        n0, b0, tot0, nrec0, fl0, slot = run(path, 0, 1 << 21)
        assert n0 >= 0
        n1, b1, tot1, nrec1, fl1, _ = run(path, 1, 1 << 21)
        assert (n1, b1, tot1, nrec1, fl1) == (n0, b0, tot0, nrec0, fl0), path
        if not str(path).endswith("multi.fa.gz"):
            assert slot >= n0, path  # the planned slot always holds a single-member file
        if n0 > 4:  # too-small slot: reports the need, never writes past cap
            n2, b2, *_ = run(path, 1, n0 - 3)
            assert n2 >= n0 and b2 == b0[: n0 - 3], path  # need is an upper bound (CRs past cap are counted)
    assert run(multi, 0, 1 << 21)[5] < run(multi, 0, 1 << 21)[0]  # exercises the CLI's retry round
    assert run("/nonexistent/x.fa", 1, 16)[0] == -1


def _unpack_py(packed, n, runs):
    """rtc_unpack_bases_dev restated: "ACGT"[code] per base, 'N' over the runs."""
    p = np.frombuffer(packed, dtype=np.uint8)
    codes = np.stack([(p >> (2 * b)) & 3 for b in range(4)], axis=1).reshape(-1)[:n]
    out = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
    for st, ln in zip(runs[0::2], runs[1::2]):
        out[st:st + ln] = ord("N")
    return out.tobytes()


def _normalised(stream):
    """what the sketch kernels make of a byte stream: ACGT upper-cased, everything else ends k-mers like 'N'"""
    a = np.frombuffer(stream, dtype=np.uint8).copy()
    up = a & 0xDF
    ok = np.isin(up, np.frombuffer(b"ACGT", dtype=np.uint8))
    return np.where(ok, up, ord("N")).astype(np.uint8).tobytes()


def test_packed_staging_equals_the_flat_stream(host, tmp_path):
    """The 2-bit packer the command lines parse into (read_genome_file_packed / PackedSink, AVX2 and portable
    paths): unpacked again with the runs it lists, the stream equals the flat reader's with every character outside
    ACGT normalised to 'N' -- for every golden file, long lines, CRLF, gzip, N runs of every alignment, lower case,
    a slot that is too small; runs are ascending, merged and inside the stream."""
    import glob
    import gzip
    host.rtch_genome_bases.restype = C.c_long
    host.rtch_genome_bases.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    host.rtch_pack_bases.restype = C.c_long
    host.rtch_pack_bases.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long]
    host.rtch_read_genome_packed.restype = C.c_int
    host.rtch_read_genome_packed.argtypes = [C.c_char_p, C.c_void_p, C.c_long, C.POINTER(C.c_long), C.c_void_p, C.c_long,
                                             C.POINTER(C.c_long), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]

    def flat(path):
        buf = np.zeros(1 << 21, dtype=np.uint8)
        tot, nrec, fl, slot = C.c_uint64(), C.c_uint64(), C.c_int(), C.c_uint64()
        n = host.rtch_genome_bases(str(path).encode(), 1, buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(tot), C.byref(nrec), C.byref(fl), C.byref(slot))
        return bytes(buf[:n]), tot.value, nrec.value

    def packed(path, cap):
        out = np.zeros(cap // 4 + 8, dtype=np.uint8)
        runs = np.zeros(1 << 18, dtype=np.uint64)
        used, nruns, tot, nrec = C.c_long(), C.c_long(), C.c_uint64(), C.c_uint64()
        st = host.rtch_read_genome_packed(str(path).encode(), out.ctypes.data_as(C.c_void_p), cap, C.byref(used), runs.ctypes.data_as(C.c_void_p),
                                          len(runs), C.byref(nruns), C.byref(tot), C.byref(nrec))
        return st, used.value, out.tobytes(), runs[:nruns.value].astype(np.int64).tolist(), tot.value, nrec.value

    rng = np.random.default_rng(5)
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=700_000)
    for st in rng.integers(0, len(seq) - 200, size=300):     # N runs of every length and alignment
        seq[st:st + int(rng.integers(1, 70))] = ord("N")
    seq[rng.random(len(seq)) < 0.2] |= 0x20                    # lower case
    seq[rng.integers(0, len(seq), size=50)] = rng.choice(np.frombuffer(b"RYKMswbdhv-*", dtype=np.uint8), size=50)
    seq = seq.tobytes()
    big = tmp_path / "big.fa"
    with open(big, "wb") as f:
        f.write(b">r1 first\r\n" + seq[:300_001] + b"\r\n\n>r2\n" + seq[300_001:300_070] + b"\n" + seq[300_070:])
    biggz = tmp_path / "big.fa.gz"
    with gzip.GzipFile(biggz, "wb", mtime=0) as f:
        f.write(open(big, "rb").read())
    lines = tmp_path / "lines.fa"   # 61-column lines: every line starts at a different offset within a packed byte
    with open(lines, "wb") as f:
        f.write(b">x\n" + b"\n".join(seq[i:i + 61] for i in range(0, 200_000, 61)) + b"\n")
    # the reader's whole-line fast path (GzStream::body_lines / PackedSink::take_lines): line lengths 1 .. 130 in turn
    # over 1.5 MB (every offset of a '\n' against the 256 KiB read buffer and the 32-byte blocks; 96 and more characters
    # take the general code), blank lines, a last line without '\n', CR LF in the middle of a file, and lines that
    # begin a record in the middle of the sequence ('@', '>')
    wide = tmp_path / "wide.fa"
    with open(wide, "wb") as f:
        f.write(b">w wide\n")
        at, n = 0, 0
        while at < 1_500_000:
            ln = 1 + n % 130
            f.write(seq[at % 600_000:at % 600_000 + ln] + (b"\n\n" if n % 97 == 0 else b"\n"))
            at += ln; n += 1
        f.write(seq[:50])
    mixed = tmp_path / "mixed.fa"
    with open(mixed, "wb") as f:
        body = b"\n".join(seq[i:i + 80] for i in range(0, 400_000, 80))
        f.write(b">m1\n" + body[:100_000] + b"\r\n" + seq[:70] + b"\r\n" + body[100_000:200_000] + b"\n@m2 second\n" + body[200_000:300_000]
                + b"\n>m3\n" + body[300_000:] + b"\n")
    # '>', '@', '+' INSIDE lines are sequence characters (kseq looks at the first character of a line only): at every offset
    # against the 32-byte blocks, so that some open a block in the middle of a line; FASTQ with reads longer than a block and
    # quality lines that begin with '@', '+' and '>'
    inner = tmp_path / "inner.fa"
    with open(inner, "wb") as f:
        f.write(b">i inner\n")
        for n in range(600):
            a = 3 + (n * 7) % 90
            f.write(seq[n * 100:n * 100 + a] + b">@+"[n % 3:n % 3 + 1] + seq[n * 100 + a:n * 100 + a + 20 + n % 40] + b"\n")
        for n in range(900):  # plain lines in front, so that the blocks run on across the line start before the special character
            for j in range(3):
                f.write(seq[n * 90 + 30 * j:n * 90 + 30 * j + 17 + (n + 5 * j) % 23] + b"\n")
            a = 1 + n % 31
            f.write(seq[n * 90:n * 90 + a] + b">@+"[n % 3:n % 3 + 1] + seq[n * 90 + a:n * 90 + a + 25] + b"\n")
    reads = tmp_path / "reads.fq"
    with open(reads, "wb") as f:
        for n in range(300):
            ln = 40 + (n * 13) % 200
            r = seq[n * 300:n * 300 + ln].replace(b"-", b"N").replace(b"*", b"N")
            q = (b"@+>"[n % 3:n % 3 + 1] + b"I" * (ln - 1)) if n % 2 else b"F" * ln
            f.write(b"@read%d x\n" % n + r + b"\n+\n" + q + b"\n")
    files = sorted(glob.glob(os.path.join(GOLD, "fasta", "*"))) + [big, biggz, lines, wide, mixed, inner, reads]
    tiers = iter((2, 1))
    for path in files + [None] + files + [None] + files:   # every SIMD tier: the CPU's best, at most AVX2, the portable loops
        if path is None:
            host.rtch_pack_force_portable(next(tiers))
            continue
        want, tot0, nrec0 = flat(path)
        st, used, pk, runs, tot, nrec = packed(path, 1 << 21)
        assert st == 0 and used == len(want) and (tot, nrec) == (tot0, nrec0), path
        assert _unpack_py(pk, used, runs) == _normalised(want), path
        starts, lens = runs[0::2], runs[1::2]
        assert all(l > 0 for l in lens) and all(a + l < b for a, l, b in zip(starts, lens, starts[1:])) and (not runs or starts[-1] + lens[-1] <= used), path
        if used > 64:  # too-small slot: status 2 and the need, nothing written past the capacity
            cap = (used - 9) // 4 * 4
            st2, used2, pk2, runs2, *_ = packed(path, cap)
            assert st2 == 2 and used2 >= used and pk2[cap // 4:] == bytes(len(pk2) - cap // 4), path
    # the bare packer on a buffer (the byte loop and the group loop meet at every offset)
    for n in (0, 1, 3, 4, 5, 31, 32, 33, 63, 64, 65, 1000, 4099):
        src = seq[7:7 + n]
        out = np.zeros(n // 4 + 8, dtype=np.uint8)
        runs = np.zeros(4096, dtype=np.uint64)
        nr = host.rtch_pack_bases(src, n, out.ctypes.data_as(C.c_void_p), runs.ctypes.data_as(C.c_void_p), len(runs))
        assert _unpack_py(out.tobytes(), n, runs[:nr].astype(np.int64).tolist()) == _normalised(src), n
    host.rtch_pack_force_portable(0)


def _py_dendrogram(n, edges, names):
    """get_newick_tree / get_linkage_from_mst restated (src/MST.cpp:1090-1150, :1246-1287): Kruskal-order
    merges with the reference's union-by-rank DSU; distinct weights, so the sort order is unambiguous."""
    edges = sorted(edges, key=lambda e: e[2])
    p, r = list(range(n)), [0] * n

    def find(x):
        while p[x] != x:
            x = p[x]
        return x

    def unite(a, b):
        a, b = find(a), find(b)
        if r[a] < r[b]:
            a, b = b, a
        p[b] = a
        if r[a] == r[b]:
            r[a] += 1
        return a
    children, height, rep = {}, {i: 0.0 for i in range(n)}, {i: i for i in range(n)}
    cid, csize, link, nxt = {i: i for i in range(n)}, {i: 1 for i in range(n)}, [], n
    for u, v, w in edges:
        ru, rv = find(u), find(v)
        if ru == rv:
            continue
        nu, nv = rep[ru], rep[rv]
        children[nxt] = [(nu, max(0.0, w - height[nu])), (nv, max(0.0, w - height[nv]))]
        height[nxt] = w
        link.append("%d\t%d\t%.6f\t%d" % (cid[ru], cid[rv], w, csize[cid[ru]] + csize[cid[rv]]))
        csize[nxt] = csize[cid[ru]] + csize[cid[rv]]
        root = unite(ru, rv)
        rep[root], cid[root] = nxt, nxt
        nxt += 1

    def build(node):
        if node not in children:
            return names[node]
        return "(" + ",".join(build(c) + ":" + "%f" % bl for c, bl in children[node]) + ")"
    return build(rep[find(0)]) + ";", link


def test_tree_and_linkage_writers_from_premsted_folder(tmp_path):
    """clust-mst --premsted DIR --newick-tree --phylip-tree --nexus-tree --linkage-matrix (no GPU involved):
    the four files against a Python restatement of the reference's dendrogram construction."""
    import subprocess
    binp = os.path.join(ROOT, "rabbittclust_amd", "bin", "clust-mst")
    if not os.path.exists(binp):
        pytest.fail("clust-mst missing: run __graft_entry__.build()")
    src = tmp_path / "src"
    src.mkdir()
    n = 40
    g = [(f"/data/it's_{i}.fna", f"seq{i}", "noName", 1000 + i) for i in range(n)]
    _write_info(src / "info.mst", g)
    rng = np.random.default_rng(3)
    w = rng.permutation(200)[: n - 1] / 997.0  # distinct weights
    edges = [(i, int(rng.integers(0, i)), float(w[i - 1])) for i in range(1, n)]  # a random spanning tree
    with open(src / "edge.mst", "wb") as f:
        f.write(struct.pack("<Q", len(edges)))
        for a, b, d in edges:
            f.write(struct.pack("<iid", a, b, d))
    out = tmp_path / "res.out"
    r = subprocess.run([binp, "--premsted", str(src), "-d", "0.05", "-o", str(out), "--newick-tree", "--phylip-tree", "--nexus-tree",
                        "--linkage-matrix"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    names = [x[0] for x in g]
    tree, link = _py_dendrogram(n, edges, names)
    assert (tmp_path / "res.out.newick.tree").read_text() == tree + "\n"
    assert (tmp_path / "res.out.phylip.tree").read_text() == "1\n" + tree + "\n"
    nexus = (tmp_path / "res.out.nexus.tree").read_text()
    assert nexus.startswith("#NEXUS\nBEGIN TAXA;\n  DIMENSIONS NTAX=40;\n  TAXLABELS '/data/it''s_0.fna'")
    assert nexus.endswith(";\nEND;\nBEGIN TREES;\n  TREE tree_1 = [&R] " + tree + "\nEND;\n")
    assert (tmp_path / "res.out.linkage.txt").read_text().splitlines() == link and len(link) == n - 1
    assert tree.count("(") == n - 1 and all(nm in tree for nm in names)
    # a forest: only the component of genome 0 is written (reference behaviour), the linkage covers every merge
    with open(src / "edge.mst", "wb") as f:
        f.write(struct.pack("<Q", len(edges) - 1))
        for a, b, d in edges[:-1]:
            f.write(struct.pack("<iid", a, b, d))
    r = subprocess.run([binp, "--premsted", str(src), "-d", "0.05", "-o", str(out), "--newick-tree", "--linkage-matrix"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    tree2, link2 = _py_dendrogram(n, edges[:-1], names)
    assert (tmp_path / "res.out.newick.tree").read_text() == tree2 + "\n" and names[n - 1] not in tree2
    assert (tmp_path / "res.out.linkage.txt").read_text().splitlines() == link2 and len(link2) == n - 2


def _write_kssd_repdb(path, thr, k, half_k, half_subk, drlevel, reps, clusters, genomes):
    """reps: [(genome id, name, length, u32 hashes)], genomes: [(name, length)] -- KssdClusterState::save_repdb
    (src/greedy.cpp:2351-2428)."""
    index = {}
    with open(path, "wb") as f:
        f.write(b"REPDB002" + struct.pack("<diiiii", thr, k, half_k, half_subk, drlevel, len(genomes)))
        f.write(struct.pack("<Q", len(reps)))
        for r, (gid, name, length, hashes) in enumerate(reps):
            f.write(struct.pack("<iiQ?I", gid, gid, length, False, len(hashes)) + struct.pack("<QQ", len(hashes), 0))
            f.write(np.asarray(hashes, dtype=np.uint32).tobytes())
            f.write(struct.pack("<Q", len(name)) + name.encode())
            for h in hashes:
                index.setdefault(int(h), []).append(r)
        f.write(struct.pack("<Q", len(clusters)))
        for c in clusters:
            f.write(struct.pack("<Q", len(c)) + struct.pack("<%di" % len(c), *c))
        f.write(struct.pack("<Q", len(genomes)))
        for name, length in genomes:
            f.write(struct.pack("<Q", len(name)) + name.encode() + struct.pack("<Q", length))
        f.write(struct.pack("<Q", len(index)))
        for h, lst in index.items():
            f.write(struct.pack("<QQ", h, len(lst)) + struct.pack("<%di" % len(lst), *lst))
    return index


def test_repdb_stats_reads_reference_layout(tmp_path):
    """clust-greedy [--fast] --db FILE --stats needs no GPU: a REPDB002 / MHREPDB1 file laid out by this test as the
    reference's save_repdb lays it out (src/greedy.cpp:2351-2428, :2789-2862) is read and reported as print_stats
    reports it (:2656-2765, :3057-3147); truncated files and wrong magics are refused."""
    import subprocess
    binp = os.path.join(ROOT, "rabbittclust_amd", "bin", "clust-greedy")
    if not os.path.exists(binp):
        pytest.fail("clust-greedy missing: run __graft_entry__.build()")
    rng = np.random.default_rng(11)
    genomes = [("/g/%d.fna" % i, 1_000_000 + 1000 * i) for i in range(9)]
    clusters = [[0, 3, 4, 7], [1], [2, 5, 6, 8]]
    reps = []
    for c in clusters:
        n = int(rng.integers(300, 600))
        hashes = np.unique(rng.integers(0, 5000, size=n).astype(np.uint32))  # small range: shared hashes between representatives
        reps.append((c[0], genomes[c[0]][0], genomes[c[0]][1], hashes))
    db = str(tmp_path / "rep.db")
    index = _write_kssd_repdb(db, 0.05, 20, 10, 6, 3, reps, clusters, genomes)
    r = subprocess.run([binp, "--fast", "--db", db, "--stats"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    postings = sum(len(v) for v in index.values())
    rep_len = sum(x[2] for x in reps)
    tot_len = sum(x[1] for x in genomes)
    want = "\n".join([
        "========================================", "        RepDB Statistics Report", "========================================", "",
        "[Basic Info]", "  Threshold:              0.05", "  Kmer size:              20", "  KSSD half_k:            10",
        "  KSSD half_subk:         6", "  KSSD drlevel:           3", "",
        "[Scale]", "  Total genomes:          9", "  Representatives:        3", "  Clusters:               3",
        "  Compression ratio:      66.67%", "",
        "[Inverted Index]", "  Unique hashes:          %d" % len(index), "  Total postings:         %d" % postings,
        "  Avg posting length:     %.2f" % (postings / len(index)), "  Max posting length:     %d" % max(len(v) for v in index.values()), "",
        "[Cluster Size Distribution]", "  Min cluster size:       1", "  Max cluster size:       4", "  Mean cluster size:      3.00",
        "  Median cluster size:    4", "  Singletons:             1 (33.3%)", "  P90 cluster size:       4", "  P95 cluster size:       4",
        "  P99 cluster size:       4", "",
        "[Representative Sketch Sizes]", "  Min sketch size:        %d" % min(len(x[3]) for x in reps),
        "  Max sketch size:        %d" % max(len(x[3]) for x in reps), "  Mean sketch size:       %.1f" % (sum(len(x[3]) for x in reps) / 3), "",
        "[Genome Coverage]", "  Total sequence length:  %d bp" % tot_len, "  Representative seq len: %d bp" % rep_len,
        "  Coverage ratio:         %.2f%%" % (100.0 * rep_len / tot_len), "========================================", ""])
    assert r.stdout == want
    assert "RepDB loaded from: " + db in r.stderr and "  Inverted index:  %d unique hashes" % len(index) in r.stderr
    raw = open(db, "rb").read()
    for cut in (4, 30, 60, len(raw) // 2, len(raw) - 3):
        bad = str(tmp_path / ("cut%d.db" % cut))
        open(bad, "wb").write(raw[:cut])
        r = subprocess.run([binp, "--fast", "--db", bad, "--stats"], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "ERROR" in r.stderr and r.stdout == ""
    r = subprocess.run([binp, "--db", db, "--stats"], capture_output=True, text=True, timeout=60)  # KSSD file read as MinHash RepDB
    assert r.returncode != 0 and "Invalid MinHash RepDB file (bad magic)" in r.stderr
    # MHREPDB1 (src/greedy.cpp:2789-2862)
    mh = str(tmp_path / "mh.db")
    with open(mh, "wb") as f:
        f.write(b"MHREPDB1" + struct.pack("<dii?", 0.03, 21, 1000, False) + struct.pack("<Q", len(reps)))
        for gid, name, length, hashes in reps:
            f.write(struct.pack("<iiQ?", gid, gid, length, False) + struct.pack("<Q", len(hashes)))
            f.write(np.asarray(hashes, dtype=np.uint64).tobytes() + struct.pack("<Q", len(name)) + name.encode())
        f.write(struct.pack("<Q", len(clusters)))
        for c in clusters:
            f.write(struct.pack("<Q", len(c)) + struct.pack("<%di" % len(c), *c))
        f.write(struct.pack("<Q", len(genomes)))
        for name, length in genomes:
            f.write(struct.pack("<Q", len(name)) + name.encode() + struct.pack("<Q", length))
        f.write(struct.pack("<Q", len(index)))
        for h, lst in index.items():
            f.write(struct.pack("<QQ", h, len(lst)) + struct.pack("<%di" % len(lst), *lst))
    r = subprocess.run([binp, "--db", mh, "--stats"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("========================================\n    MinHash RepDB Statistics Report\n")
    assert "  Threshold:              0.03\n  Kmer size:              21\n  Sketch size:            1000\n  Containment mode:       no\n\n[Scale]" in r.stdout
    assert "[Representative Sketch Sizes]" not in r.stdout
    assert "  P99 cluster size:       4\n\n[Genome Coverage]\n" in r.stdout
    assert "  Unique hashes:          %d\n" % len(index) in r.stdout
    r = subprocess.run([binp, "--fast", "--db", mh, "--stats"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "Invalid RepDB file (bad magic)" in r.stderr
