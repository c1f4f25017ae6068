"""KSSD sketches of one random genome through the bucket-index kernel, the cuckoo-index kernel and the oracle,
for several k: prints the set differences (all zero when the kernels agree with the restatement)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rabbittclust_amd import api, host
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import pyoracle as oracle
rng = np.random.default_rng(1)
L = 300000
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
off = np.array([0, L], dtype=np.uint64)
sd = host.generate_shuffle_dim(6)
ctxs = {}
for name, env in (("bucket", None), ("cuckoo", "1")):
    if env: os.environ["RTC_KSSD_CUCKOO"] = env
    ctxs[name] = api.Context(0)
for k in (21, 19, 17, 27, 15, 23):
    want = oracle.kssd_sketch(seq, k, 3)
    for name, ctx in ctxs.items():
        if name == "cuckoo": os.environ["RTC_KSSD_CUCKOO"] = "1"
        else: os.environ.pop("RTC_KSSD_CUCKOO", None)
        d = ctx.upload_sequences(seq)
        sk = ctx.sketch_kssd(d, off, sd, kmer_size=k, drlevel=3); ctx.sync()
        a = sk.to_host()[0]
        ex = np.setdiff1d(a, want); mi = np.setdiff1d(want, a)
        print(k, name, "got", len(a), "want", len(want), "extra", len(ex), "missing", len(mi), [hex(int(v)) for v in ex[:4]], [hex(int(v)) for v in mi[:4]], flush=True)
