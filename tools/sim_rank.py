"""Per-rank work of the multi-GPU bench shape on ONE GPU: n_total sketches, rank r of `world`.
Usage: [SIM_KSSD=1] sim_rank.py [n_total] [world] [rank] [length]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api, pipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else world - 1
L = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000
ctx = api.Context(0)
KSSD = os.environ.get("SIM_KSSD") == "1"   # BASELINE config 5's shape: n_total KSSD sketches (--fast, u32) of `length`-base genomes
if KSSD:
    from rabbittclust_amd import host
    from rabbittclust_amd.api import SketchSet
    sd = host.generate_shuffle_dim(6)
    per = n // world                          # sketched a rank's share at a time (the genomes of all ranks do not fit at once)
    stride = int(L / 4096 * 1.5) + 256
    rows, lens = [], []
    for r in range(world):
        desc = api.synth_family_descs(per // 10, 10, global_seed=42 + 1000 * r)
        off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
        seq = ctx.synth_genomes(desc, off)
        part = ctx.sketch_kssd(seq, off, sd, kmer_size=21, drlevel=3, stride=stride)
        ctx.sync()
        rows.append(part.hashes.view(part.n, -1)); lens.append(part.len)
        del seq
    hashes = torch.cat(rows).contiguous(); lens = torch.cat(lens).contiguous()
    start = torch.arange(hashes.shape[0], dtype=torch.int64, device=hashes.device) * hashes.shape[1]
    sk = SketchSet(hashes.view(-1), start, lens, 4, 22, "kssd")
    del rows
else:
    desc = api.synth_family_descs(n // 10, 10, global_seed=42)
    off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sk = ctx.sketch_minhash(seq, off, k=21, size=1000)
    ctx.sync()
    del seq
fixed = float(sys.argv[5]) if len(sys.argv) > 5 else 1.84 * float(sk.len.float().mean().item())
b = pipeline.triangle_row_ranges(sk.n, world, fixed_cols=fixed)
pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=1000, threshold=0.05, rank=rank, world=world, mode="kssd" if KSSD else "minhash")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    edges, m = pipe.candidate_edges(sk, b[rank], b[rank + 1])
    torch.cuda.synchronize(); t1 = time.time()
    sel, rounds = pipe.boruvka(sk, edges, m)
    torch.cuda.synchronize(); t2 = time.time()
    rows = b[rank + 1] - b[rank]
    pairs = (b[rank + 1] * (b[rank + 1] - 1) - b[rank] * (b[rank] - 1)) // 2
    print(f"n={sk.n} rank {rank}/{world}: rows {rows}, pairs {pairs:.3e}, cand edges {m}, pair phase {1e3*(t1-t0):.1f} ms "
          f"({pairs/(t1-t0):.3e} pairs/s), local boruvka {1e3*(t2-t1):.1f} ms ({rounds} rounds, {len(sel)} forest edges)", flush=True)
