"""Host-only parse rate of the command lines' FASTA reader (read_genome_file_packed) over plain files in tmpfs.
Usage: parse_bench.py [n_files] [length] [threads]
Measured with it in round 3: mmap(MAP_POPULATE) of the plain file instead of read(2) into a 256 KiB buffer is 25 % SLOWER
(tmpfs, 8 threads: 13 against 18 GB/s) -- the reader keeps read(2)."""
import ctypes as C, os, sys, time, tempfile, shutil
from concurrent.futures import ThreadPoolExecutor
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(root, "rabbittclust_amd", "librtclust_host.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tmp = tempfile.mkdtemp(prefix="rtc_pb_", dir="/dev/shm")
rng = np.random.default_rng(1)
base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L)]
nl = np.full((L // 80, 1), 10, dtype=np.uint8)
body = np.concatenate([base.reshape(-1, 80), nl], axis=1).tobytes()
paths = []
for g in range(n):
    p = os.path.join(tmp, f"g{g}.fna")
    open(p, "wb").write(f">g{g} x\n".encode() + body)
    paths.append(p.encode())
cap = (L + 4096) // 64 * 64
lib.rtch_read_genome_packed.restype = C.c_int
def work(tid):
    out = (C.c_ubyte * (cap // 4 + 64))()
    runs = (C.c_ulonglong * 4096)()
    used, nruns = C.c_long(), C.c_long()
    tot, nrec = C.c_ulonglong(), C.c_ulonglong()
    for i in range(tid, n, T):
        st = lib.rtch_read_genome_packed(paths[i], out, C.c_long(cap), C.byref(used), runs, C.c_long(2048), C.byref(nruns), C.byref(tot), C.byref(nrec))
        assert st == 0 and tot.value == L, (st, tot.value)
for rep in range(3):
    t0 = time.time()
    with ThreadPoolExecutor(T) as ex:
        list(ex.map(work, range(T)))
    dt = time.time() - t0
    print(f"{n} files x {L} bp, {T} threads: {dt*1e3:.1f} ms = {n*L/dt/1e9:.2f} GB/s")
shutil.rmtree(tmp)
