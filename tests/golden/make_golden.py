"""Generates the committed fixtures under tests/golden/.  Run in the build container:

    python tests/golden/make_golden.py

Two kinds of fixtures, labelled in MANIFEST.json:
  * "reference": outputs of the reference's own code run here -- oracle/_ref/libref_harness.so is
    compiled from /root/reference/src/{kseq.h,UnionFind.h} where they lie (oracle/Makefile).
    They pin the host FASTA reader and the union-find used by Kruskal.
  * "oracle": outputs of oracle/ (the CPU restatement).  They let the GPU box check the HIP path
    without regenerating expectations, and freeze the oracle against accidental edits.  The MinHash
    ones are NOT pinned against upstream RabbitSketch (see oracle/rtc_oracle.h).
"""
import ctypes as C
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")


def write_fasta_inputs():
    rng = np.random.default_rng(2024)

    def seq(n):
        return "".join(rng.choice(list("ACGT"), size=n))
    files = {}
    s1 = seq(500)
    files["plain.fa"] = ">chr1 first record comment\n" + "\n".join(s1[i:i + 80] for i in range(0, 500, 80)) + "\n" \
        ">chr2\n" + seq(130) + "\n>chr3\tTabbed comment here\n" + seq(77) + "\n"
    files["crlf_blank.fa"] = ">a desc\r\n" + seq(60) + "\r\n\r\n" + seq(45) + "\r\n>b\r\nACGTNNNNacgtRYK\r\n"
    files["noeol.fa"] = ">only_name\n" + seq(100)
    files["leading_junk.fa"] = "junk line before any header\n\n>x  two  spaces\nAC\nGT\n\n>y z\n\n>z\nTTTT"
    files["reads.fq"] = "@r1 fastq comment\nACGTACGTAC\n+\nIIIIIIIIII\n@r2\nGGGGCCCC\nTT\n+r2\nIIIIIIII\nII\n"
    files["empty.fa"] = ""
    for name, text in files.items():
        with open(os.path.join(HERE, "fasta", name), "w", newline="") as f:
            f.write(text)
    with open(os.path.join(HERE, "fasta", "plain.fa.gz"), "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
            f.write(files["plain.fa"].encode())
    return sorted(list(files) + ["plain.fa.gz"])


def main():
    manifest = {}
    names = write_fasta_inputs()
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/libref_harness.so missing: run `make -C oracle` where /root/reference exists")
    ref = C.CDLL(REF)
    ref.ref_kseq_dump.restype = C.c_long
    for n in names:
        p = os.path.join(HERE, "fasta", n).encode()
        need = ref.ref_kseq_dump(p, None, 0)
        buf = C.create_string_buffer(max(need, 1))
        ref.ref_kseq_dump(p, buf, need)
        with open(os.path.join(HERE, "kseq_dump_" + n + ".txt"), "wb") as f:
            f.write(buf.raw[:need])
        manifest["kseq_dump_" + n + ".txt"] = {"kind": "reference", "source": "kseq.h via oracle/ref_harness.cpp"}
    # union-find roots after a fixed merge sequence
    rng = np.random.default_rng(7)
    n, m = 200, 150
    xs = rng.integers(0, n, size=m).astype(np.int32)
    ys = rng.integers(0, n, size=m).astype(np.int32)
    roots = np.zeros(n, dtype=np.int32)
    size = C.c_int()
    ref.ref_unionfind(n, xs.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), m,
                      roots.ctypes.data_as(C.c_void_p), C.byref(size))
    np.savez(os.path.join(HERE, "unionfind.npz"), xs=xs, ys=ys, roots=roots, size=size.value)
    manifest["unionfind.npz"] = {"kind": "reference", "source": "UnionFind.h via oracle/ref_harness.cpp"}

    # ---- oracle fixtures ----
    sd = O.kssd_shuffle_dim(6)
    kept = np.nonzero(sd < 4096)[0].astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "kssd_shuffle_hs6.npz"), dim_id=kept, rank=sd[kept].astype(np.uint16),
                        head=sd[:64].astype(np.int32))
    manifest["kssd_shuffle_hs6.npz"] = {"kind": "oracle", "source": "glibc srand/rand via orc_kssd_shuffle_dim(6); "
                                        "4096 surviving (dim_id, rank) pairs + first 64 table entries"}
    L = 60_000
    descs = [(11, 0, 0, 0), (11, 5, 300, 0), (11, 6, 900, 0), (12, 0, 0, 0), (12, 9, 500, 7000), (13, 0, 0, 0)]
    genomes = [O.synth_genome(f, m_, t, L, ne) for (f, m_, t, ne) in descs]
    seq = np.concatenate(genomes)
    off = np.arange(len(genomes) + 1, dtype=np.uint64) * L
    mh = O.sketch_minhash_batch(seq, off, 21, 400, threads=1)
    ks = [O.kssd_sketch(g, 21, 3) for g in genomes]
    flat, start, lens = O.to_csr(mh)
    mst = O.mst(flat, start, lens, 21, 0, 0.05, threads=1)
    np.savez_compressed(os.path.join(HERE, "sketch_fixture.npz"), descs=np.array(descs, dtype=np.uint64), L=L,
                        minhash=np.array(mh, dtype=object), kssd=np.array(ks, dtype=object),
                        mst=mst, allow_pickle=True)
    manifest["sketch_fixture.npz"] = {"kind": "oracle", "source": "6 synthetic 60 kbp genomes: MinHash k=21 s=400 "
                                      "(parity unpinned vs RabbitSketch), KSSD k=21 dr=3, MST at d=0.05"}
    vec = []
    for key, seed in [(b"", 0), (b"A", 42), (b"ACGTACGTACGTACGTACGTA", 42), (b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", 42),
                      (bytes(range(40)), 7)]:
        h1, h2 = O.murmur3_x64_128(key, seed)
        vec.append({"key_hex": key.hex(), "seed": seed, "h1": h1, "h2": h2})
    with open(os.path.join(HERE, "murmur3_vectors.json"), "w") as f:
        json.dump({"smhasher_verification": hex(O.lib().orc_murmur3_smhasher_verification()), "vectors": vec}, f, indent=1)
    manifest["murmur3_vectors.json"] = {"kind": "oracle", "source": "orc_murmur3_x64_128, itself pinned by the public "
                                        "SMHasher verification value 0x6384BA69"}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest), "fixtures")


if __name__ == "__main__":
    main()
