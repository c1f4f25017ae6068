"""Same-process A/B of the two MinHash sketch kernels on one synthetic batch: ASCII input (rtc_sketch_minhash_dev) against the
2-bit staging format (rtc_sketch_minhash_packed_dev), alternating, and a check that the sketches are identical.
Usage: python tools/run_sketch_packed.py [n] [length] [reps] [size] [k] [only: ascii|packed|both] [n_every: an 8-base run of N every so many bases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
size = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
k = int(sys.argv[5]) if len(sys.argv) > 5 else 21
only = sys.argv[6] if len(sys.argv) > 6 else "both"
n_every = int(sys.argv[7]) if len(sys.argv) > 7 else 0
ctx = api.Context(0)
desc = api.synth_family_descs(max(1, n // 10), 10, global_seed=42, n_every=n_every)
n = len(desc)
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
seq = ctx.synth_genomes(desc, off); ctx.sync()
pb = api.pack_staging(seq, int(off[-1])); ctx.sync()
ref = None
for r in range(reps):
    for which in ("ascii", "packed"):
        if only not in ("both", which):
            continue
        ctx.timer_start()
        sk = ctx.sketch_minhash(seq, off, k=k, size=size) if which == "ascii" else ctx.sketch_minhash_packed(pb, off, k=k, size=size)
        ms = ctx.timer_stop()
        print(f"{which:6s} {n} x {L} k={k} s={size}" + (f" N/{n_every}" if n_every else "") + f": {ms:.2f} ms  {n*L/ms/1e6:.1f} Gbp/s", flush=True)
        if ref is None:
            ref = sk
        elif r == 0:
            same = torch.equal(ref.hashes, sk.hashes) and torch.equal(ref.len, sk.len)
            print("identical sketches:", same, flush=True)
            if not same:
                sys.exit(1)
