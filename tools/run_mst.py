"""Times the MST phase of the two bench shapes split into its parts: Boruvka rounds on the device, read-back, host
distances + sort.  Usage: run_mst.py minhash|kssd [n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api, pipeline, host
mode = sys.argv[1] if len(sys.argv) > 1 else "minhash"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (10000 if mode == "minhash" else 25000)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = api.Context(0)
desc = api.synth_family_descs(n // 10, 10, global_seed=42)
L = 1_000_000 if mode == "minhash" else 2_000_000
off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
seq = ctx.synth_genomes(desc, off)
sk = ctx.sketch_minhash(seq, off, k=21, size=1000) if mode == "minhash" else ctx.sketch_kssd(seq, off, host.generate_shuffle_dim(6), kmer_size=21, drlevel=3)
ctx.sync(); del seq
pipe = pipeline.MstPipeline(ctx, k=sk.k, threshold=0.05)
edges, m = pipe.candidate_edges(sk, 0, sk.n)
for it in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sel, rounds = pipe.boruvka(sk, edges, m)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mst = pipe.finish(sk, sel)
    t2 = time.perf_counter()
    print(f"{mode} n={sk.n} m={m}: boruvka ({rounds} rounds, incl. forest read-back) {1e3*(t1-t0):.3f} ms, finish (lens read-back + distances + sort) {1e3*(t2-t1):.3f} ms", flush=True)
