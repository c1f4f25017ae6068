"""End-to-end timing of the clust-mst command line from FASTA files (host parse + PCIe + GPU).
Usage: cli_e2e.py [n_genomes] [length]"""
import os, subprocess, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
assert L % 80 == 0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="rtc_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
ctx = api.Context(0)
desc = api.synth_family_descs(max(1, n // 8), 8, global_seed=77)[:n]
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
seq = ctx.synth_genomes(desc, off).cpu().numpy()
t0 = time.time()
paths = []
nl = np.full((L // 80, 1), 10, dtype=np.uint8)
for g in range(n):
    p = os.path.join(tmp, f"g{g:05d}.fna")
    body = np.concatenate([seq[g * L:(g + 1) * L].reshape(-1, 80), nl], axis=1).tobytes()
    with open(p, "wb") as f:
        f.write(f">g{g} synthetic\n".encode() + body)
    paths.append(p)
open(os.path.join(tmp, "list.txt"), "w").write("\n".join(paths) + "\n")
print(f"wrote {n} FASTA files ({n * L / 1e9:.2f} Gbp) in {time.time() - t0:.1f}s", flush=True)
del ctx
for binname, extra in (("clust-mst", ["-s", "1000"]), ("clust-mst", ["--fast"]), ("clust-mst", ["-s", "1000"])):
    t0 = time.time()
    r = subprocess.run(["env", "RTC_VERBOSE=1", os.path.join(root, "rabbittclust_amd", "bin", binname), "-l", "-i", os.path.join(tmp, "list.txt"), "-k", "21",
                        "-d", "0.05", "-e", "-o", os.path.join(tmp, "out.cluster")] + extra, capture_output=True, text=True, cwd=tmp)
    dt = time.time() - t0
    lines = [ln for ln in r.stderr.splitlines() if "time of" in ln or "cluster number" in ln or ln.startswith(("[gpu", "[parse]", "[plan]", "[init]", "[free]", "[tune]", "[share]", "[mst", "[ctx]", "[exit]"))]
    print(binname, " ".join(extra), f"rc={r.returncode} wall={dt:.2f}s  {n * L / dt / 1e9:.2f} Gbp/s end-to-end from files", flush=True)
    for ln in lines:
        print("   ", ln)
    if r.returncode != 0:
        print(r.stderr[-2000:])
subprocess.run(["rm", "-rf", tmp])
