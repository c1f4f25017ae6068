"""Small driver used under rocprofv3: runs the KSSD sketch kernel a few times on synthetic genomes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k = int(sys.argv[4]) if len(sys.argv) > 4 else 21
dr = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ctx = api.Context(0)
desc = api.synth_family_descs(max(1, n // 10), 10, global_seed=42)
n = len(desc)
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
seq = ctx.synth_genomes(desc, off); ctx.sync()
sd = host.generate_shuffle_dim(6 if 6 - dr >= 2 else dr + 2)
for r in range(reps):
    ctx.timer_start()
    sk = ctx.sketch_kssd(seq, off, sd, kmer_size=k, drlevel=dr)
    ms = ctx.timer_stop()
    print(f"kssd sketch {n} x {L} k={k} drlevel={dr}: {ms:.2f} ms  {n*L/ms/1e6:.1f} Gbp/s  mean tuples {sk.len.float().mean().item():.0f}", flush=True)
