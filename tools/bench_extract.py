"""Times rtc_extract_edges_dev alone on a 10 000 x 10 000 count matrix with ~150 000 survivors."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from rabbittclust_amd import api
ctx = api.Context(0)
n = 10000
common = torch.zeros((n, n), dtype=torch.int32, device=ctx.device)
idx = torch.randint(0, n * n, (300000,), device=ctx.device)
common.view(-1)[idx] = 5
lens = torch.full((n,), 1000, dtype=torch.int32, device=ctx.device)
class SK: pass
sk = SK(); sk.len = lens; sk.n = n
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    edges, count = ctx.extract_edges(common, sk, 0, n, 0, n, 4, cap=1 << 20)
    torch.cuda.synchronize(); print(f"extract {1e3*(time.time()-t0):.3f} ms count {int(count.item())}")
