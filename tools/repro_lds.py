import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api
s = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
ctx = api.Context(0)
rng = np.random.default_rng(1)
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(sys.argv[3]) if len(sys.argv) > 3 else 400_000)
off = np.array([0, len(seq)], dtype=np.uint64)
sk = ctx.sketch_minhash(ctx.upload_sequences(seq), off, k=k, size=s)
ctx.sync()
print("ok", s, k, int(sk.len[0]))
