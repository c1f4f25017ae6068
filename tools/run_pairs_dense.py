"""Pair phase on a DENSE input: `fam` families of n/fam near-identical sketches (every pair of a family shares most of its
hashes, posting lists as long as a family) -- the regime the tiled kernel keeps (the inverted join would walk a
quadratic number of co-occurrences).  Default dispatch: the cost rule itself has to pick the tiled kernel.
Usage: run_pairs_dense.py [n] [max_rate] [reps] [families] [minhash|kssd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api, pipeline, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
fam = int(sys.argv[4]) if len(sys.argv) > 4 else 10
mode = sys.argv[5] if len(sys.argv) > 5 else "minhash"
ctx = api.Context(0)
desc = api.synth_family_descs(fam, n // fam, global_seed=42, max_rate=rate)
L = 500_000 if mode == "minhash" else 2_000_000
off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
seq = ctx.synth_genomes(desc, off)
if mode == "minhash":
    sk = ctx.sketch_minhash(seq, off, k=21, size=1000)
else:
    sk = ctx.sketch_kssd(seq, off, host.generate_shuffle_dim(6), kmer_size=21, drlevel=3)
ctx.sync(); del seq
pipe = pipeline.MstPipeline(ctx, k=sk.k, threshold=0.05)
for it in range(reps):
    ctx.timer_start()
    edges, m = pipe.candidate_edges(sk, 0, sk.n)
    ms = ctx.timer_stop()
    e = edges[:m, 2].long()
    print(f"dense {mode} n={sk.n} fam={fam} rate<={rate}: pair phase {ms:.3f} ms (path {ctx.pair_last_path()}), {m} candidate edges, "
          f"mean common {float(e.float().mean()):.1f}, {sk.n * (sk.n - 1) / 2 / ms / 1e6:.2f} Gpairs/s", flush=True)
