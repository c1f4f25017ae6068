"""Timeline of the clust-mst command line on N synthetic FASTA files in tmpfs (RTC_VERBOSE lines), several runs.
Usage: cli_timeline.py [n_genomes] [runs] [extra clust-mst args...]"""
import os, subprocess, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rabbittclust_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = sys.argv[3:] or ["-s", "1000"]
L = 5_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="rtc_tl_", dir="/dev/shm")
ctx = api.Context(0)
desc = api.synth_family_descs(max(1, n // 8), 8, global_seed=77)[:n]
nl = np.full((L // 80, 1), 10, dtype=np.uint8)
paths = []
for c0 in range(0, n, 256):
    c1 = min(n, c0 + 256)
    off = np.arange(c1 - c0 + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc[c0:c1], off).cpu().numpy()
    for g in range(c0, c1):
        p = os.path.join(tmp, f"g{g:05d}.fna")
        with open(p, "wb") as f:
            f.write(f">g{g} synthetic\n".encode() + np.concatenate([seq[(g - c0) * L:(g - c0 + 1) * L].reshape(-1, 80), nl], axis=1).tobytes())
        paths.append(p)
open(os.path.join(tmp, "list.txt"), "w").write("\n".join(paths) + "\n")
del ctx
for r in range(runs):
    time.sleep(float(os.environ.get("TL_SLEEP", "0")))  # the previous process's GPU state is torn down by the driver after it has left
    t0 = time.time()
    res = subprocess.run(["env", "RTC_VERBOSE=1"] + [f"{k}={v}" for k, v in os.environ.items() if k.startswith("RTC_")] + [ os.path.join(root, "rabbittclust_amd", "bin", "clust-mst-measure" if os.path.exists(os.path.join(root, "rabbittclust_amd", "bin", "clust-mst-measure")) else "clust-mst"), "-l", "-i", os.path.join(tmp, "list.txt"), "-k", "21",
                          "-d", "0.05", "-e", "-o", os.path.join(tmp, "out.cluster")] + extra, capture_output=True, text=True, cwd=tmp)
    dt = time.time() - t0
    import re
    m = re.search(r"GPU context\(s\) in ([0-9.]+)s", res.stderr)
    init = float(m.group(1)) if m else 0.0
    m = re.search(r"output written at t\+([0-9.]+)s", res.stderr)
    written = float(m.group(1)) if m else 0.0
    m = re.search(r"main entered at ([0-9.]+), leaving at ([0-9.]+)", res.stderr)
    start, leave = (float(m.group(1)) - t0, t0 + dt - float(m.group(2))) if m else (0.0, 0.0)
    print(f"run {r}: rc={res.returncode} wall={dt:.3f}s {n * L / dt / 1e9:.1f} Gbp/s   (process start {start:.3f}, runtime up at {init:.3f}, output "
          f"{written - init:.3f} s later, _exit to reaped {leave:.3f} s)", flush=True)
    keep = [ln for ln in res.stderr.splitlines() if ln.startswith(("[init]", "[ctx]", "[plan]", "[tune]", "[exit]", "[free]", "[share]")) or "time of" in ln]
    gp = [ln for ln in res.stderr.splitlines() if ln.startswith("[gpu")]
    for ln in res.stderr.splitlines():
        if ln.startswith("[probe]"):
            print("   ", ln)
    if os.environ.get("TL_BRIEF"):
        print("   ", " | ".join(ln.strip() for ln in keep if ln.startswith(("[init]  ", "[ctx]", "[exit]")) and "list" not in ln))
    else:
        for ln in keep + gp[:3] + (["   ..."] if len(gp) > 3 else []):
            print("   ", ln)
shutil.rmtree(tmp)
