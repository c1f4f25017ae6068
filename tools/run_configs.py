"""Timings of the BASELINE.json configs that are parity-test cases rather than the bench line
(configs 4 and 5, scaled to one GPU).  Usage: run_configs.py kssd|greedy [n] [L]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api, pipeline

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
ctx = api.Context(0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if mode == "kssd":
    desc = api.synth_family_descs(n // 10, 10, global_seed=42)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off); ctx.sync()
    host = C.CDLL(os.path.join(root, "rabbittclust_amd", "librtclust_host.so"))
    sd = np.zeros(1 << 24, dtype=np.int32); host.rtch_shuffle_dim(6, sd.ctypes.data_as(C.c_void_p))
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sk = ctx.sketch_kssd(seq, off, sd, kmer_size=21, drlevel=3)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"kssd sketch {n} x {L}: {dt*1e3:.1f} ms  {n*L/dt/1e9:.1f} Gbp/s  mean tuples {sk.len.float().mean().item():.0f}", flush=True)
    pipe = pipeline.MstPipeline(ctx, k=sk.k, threshold=0.05)
    for r in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        edges, m = pipe.candidate_edges(sk, 0, sk.n)
        sel, rounds = pipe.boruvka(sk, edges, m)
        mst = pipe.finish(sk, sel)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"kssd all-pairs+mst: {dt*1e3:.1f} ms  {n*(n-1)/2/dt/1e9:.2f} Gpairs/s  cand={m} mst={len(mst)}", flush=True)
elif mode == "greedy":
    rng = np.random.default_rng(1)
    fam = 10
    desc = api.synth_family_descs(n // fam, fam, global_seed=43, max_rate=0.04)
    frac = rng.uniform(0.2, 1.0, size=n); frac[::fam] = 1.0       # prefix genomes (containment families)
    lens = (frac * L).astype(np.uint64)
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum((lens + 15) // 16 * 16)
    # prefix genomes share the ancestor stream: same desc, shorter length
    seq = ctx.synth_genomes(desc, off); ctx.sync()
    # genome g occupies [off[g], off[g]+lens[g]); the pad bytes up to off[g+1] are whatever synth wrote (same stream) -> use true ends
    off2 = np.stack([off[:-1], off[:-1] + lens], axis=1)
    sizes = np.maximum((lens * 1.0125 / 1000).astype(np.uint32), 100)   # ~ fileBytes / 1000
    for r in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sk = ctx.sketch_minhash(seq, off, k=21, sizes=sizes)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"containment sketch {n} genomes ({off[-1]/1e9:.1f} Gbp): {dt*1e3:.1f} ms {off[-1]/dt/1e9:.1f} Gbp/s", flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ncl, rep = ctx.greedy(sk, 0.05, size_cfg=sizes, is_containment=True)
    dt = time.perf_counter() - t0
    print(f"greedy {n} genomes, sizes {sizes.min()}..{sizes.max()}: {dt:.2f} s  clusters={ncl}", flush=True)
