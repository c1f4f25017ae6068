"""Co-occurrence statistics of the bench shapes: candidate edges, sum of common (= posting-list pair count
of the reference's inverted index), posting-list length histogram.  Usage: cooc_stats.py minhash|kssd [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api, pipeline, host
mode = sys.argv[1] if len(sys.argv) > 1 else "minhash"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (10000 if mode == "minhash" else 25000)
ctx = api.Context(0)
desc = api.synth_family_descs(n // 10, 10, global_seed=42)
if mode == "minhash":
    L = 1_000_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sk = ctx.sketch_minhash(seq, off, k=21, size=1000)
else:
    L = 2_000_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sk = ctx.sketch_kssd(seq, off, host.generate_shuffle_dim(6), kmer_size=21, drlevel=3)
ctx.sync()
del seq
pipe = pipeline.MstPipeline(ctx, k=sk.k, threshold=0.05)
pipe.radio_off = True
edges, m = pipe.candidate_edges(sk, 0, sk.n)
e = edges[:m].cpu().numpy().view(np.uint32)
print(f"{mode} n={sk.n}: {m} candidate edges (radio filter on), sum common {int(e[:,2].astype(np.int64).sum())}")
lens = sk.len.cpu().numpy()
hs = sk.hashes.cpu().numpy().reshape(-1)
st = sk.start.cpu().numpy()
allk = np.concatenate([hs[int(st[g]):int(st[g]) + int(lens[g])] for g in range(sk.n)])
u, c = np.unique(allk, return_counts=True)
print(f"keys {len(allk)}, distinct {len(u)}, sum m(m-1)/2 = {int((c.astype(np.int64) * (c - 1) // 2).sum())}, max m {c.max()}")
print("posting length histogram:", np.bincount(np.minimum(c, 20))[:21])
