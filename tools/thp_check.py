"""Does this host give transparent huge pages to an madvise()d anonymous mapping, and what does leaving with 1 GiB touched cost?"""
import mmap, os, time, subprocess, sys
for f in ("enabled", "defrag", "shmem_enabled"):
    try:
        print(f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except Exception as e:
        print(f, e)
if len(sys.argv) > 1:
    n = 1 << 30
    m = mmap.mmap(-1, n + (2 << 20))
    if sys.argv[1] == "thp":
        m.madvise(14)  # MADV_HUGEPAGE
    t0 = time.time()
    for o in range(0, n, 4096):
        m[o] = 1
    t1 = time.time()
    huge = [l for l in open("/proc/self/smaps_rollup") if l.startswith("AnonHugePages")]
    print(sys.argv[1], "touch %.3f s" % (t1 - t0), huge[0].strip() if huge else "", flush=True)
    open("/tmp/t_end", "w").write(repr(time.time()))
    os._exit(0)
else:
    for mode in ("plain", "thp"):
        subprocess.run([sys.executable, __file__, mode])
        print("  exit took %.3f s" % (time.time() - float(open("/tmp/t_end").read())))
