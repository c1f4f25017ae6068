import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rabbittclust_amd import api, pipeline
ctx = api.Context(0)
def mk(n, L):
    desc = api.synth_family_descs(n // 8, 8, global_seed=1)
    off = np.arange(len(desc) + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sk = ctx.sketch_minhash(seq, off, k=21, size=1000); ctx.sync(); return sk
small, big = mk(64, 100000), mk(4096, 200000)
os.environ["RTC_PAIR_JOIN"] = "2"
def t(sk, label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = ctx.mst(sk, 0.05)
    torch.cuda.synchronize(); print(f"{label}: rtc_mst {1e3*(time.perf_counter()-t0):.2f} ms ({len(m)} edges)", flush=True)
t(small, "small, first call"); t(small, "small, second"); t(big, "big, first"); t(big, "big, second")
