"""Pair phase on a DENSE input: n sketches of ONE family (every pair shares hashes) -- the regime the tiled kernel keeps.
Usage: run_pairs_dense.py [n] [max_rate] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api, pipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.08
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = api.Context(0)
desc = api.synth_family_descs(1, n, global_seed=42, max_rate=rate)
L = 500_000
off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
seq = ctx.synth_genomes(desc, off)
sk = ctx.sketch_minhash(seq, off, k=21, size=1000)
ctx.sync(); del seq
pipe = pipeline.MstPipeline(ctx, k=sk.k, threshold=0.05)
for it in range(reps):
    ctx.timer_start()
    edges, m = pipe.candidate_edges(sk, 0, sk.n)
    ms = ctx.timer_stop()
    e = edges[:m, 2].long()
    print(f"dense n={sk.n} rate<={rate}: pair phase {ms:.3f} ms (path {ctx.pair_last_path()}), {m} candidate edges, mean common {float(e.float().mean()):.1f}, "
          f"{sk.n * (sk.n - 1) / 2 / ms / 1e6:.2f} Gpairs/s", flush=True)
