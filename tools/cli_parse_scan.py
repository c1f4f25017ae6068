"""Host-side scaling of the clust-mst FASTA pipeline: threads x (pinned | pageable) staging.
Usage: cli_parse_scan.py [n_genomes] [length]"""
import os, subprocess, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rabbittclust_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="rtc_scan_", dir="/dev/shm")
ctx = api.Context(0)
desc = api.synth_family_descs(max(1, n // 8), 8, global_seed=77)[:n]
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
seq = ctx.synth_genomes(desc, off).cpu().numpy()
nl = np.full((L // 80, 1), 10, dtype=np.uint8)
paths = []
for g in range(n):
    p = os.path.join(tmp, f"g{g:05d}.fna")
    with open(p, "wb") as f:
        f.write(f">g{g} synthetic\n".encode() + np.concatenate([seq[g * L:(g + 1) * L].reshape(-1, 80), nl], axis=1).tobytes())
    paths.append(p)
open(os.path.join(tmp, "list.txt"), "w").write("\n".join(paths) + "\n")
del ctx
for env_extra in ({}, {"RTC_STAGE_PINNED": "1"}, {"RTC_BATCH_BYTES": str(256 << 20)}, {"RTC_BATCH_BYTES": str(512 << 20)},
                  {"RTC_BATCH_BYTES": str(1 << 30)}, {"RTC_BATCH_BYTES": str(4 << 30)}):
    for t in (8, 16, 32):
        env = dict(os.environ, RTC_VERBOSE="1", **env_extra)
        t0 = time.time()
        r = subprocess.run([os.path.join(root, "rabbittclust_amd", "bin", "clust-mst"), "-l", "-i", os.path.join(tmp, "list.txt"),
                            "-k", "21", "-d", "0.05", "-e", "-t", str(t), "-o", os.path.join(tmp, "o")], capture_output=True, text=True, cwd=tmp, env=env)
        dt = time.time() - t0
        parse = [float(ln.split(" in ")[1][:-1]) for ln in r.stderr.splitlines() if ln.startswith("[parse]")]
        h2d = [float(ln.split("h2d ")[1].split("s")[0]) for ln in r.stderr.splitlines() if ln.startswith("[gpu]")]
        plan = [ln for ln in r.stderr.splitlines() if ln.startswith("[plan]")]
        print(f"{str(env_extra):36s} t={t:3d} rc={r.returncode} wall={dt:.2f}s parse={sum(parse):.2f}s ({n*L/max(sum(parse),1e-9)/1e9:.1f} GB/s) "
              f"h2d={sum(h2d):.2f}s plan={plan[0].split(',')[-1] if plan else '?'}", flush=True)
subprocess.run(["rm", "-rf", tmp])
