"""Times the pair phase (plan + pair_tiled_kernel with fused edge emission) of the two bench shapes.
Usage: run_pairs.py minhash|kssd [n] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api, pipeline, host
mode = sys.argv[1] if len(sys.argv) > 1 else "minhash"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (10000 if mode == "minhash" else 25000)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
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
for it in range(reps):
    ctx.timer_start()
    edges, m = pipe.candidate_edges(sk, 0, sk.n)
    ms = ctx.timer_stop()
    print(f"{mode} n={sk.n}: pair phase {ms:.3f} ms, {m} candidate edges, {sk.n * (sk.n - 1) / 2 / ms / 1e6:.2f} Gpairs/s", flush=True)
