"""Summarise rocprofv3 sqlite output: per-kernel durations (kernel-trace) or summed PMC counters."""
import glob, sqlite3, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    con = sqlite3.connect(f); cur = con.cursor()
    print("==", f)
    try:
        rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception:
        rows = []
    if rows:
        for r in rows:
            if not r[0].startswith("void at::"):
                print(f"  {r[0][:60]:60s} {r[1]:28s} {r[2]:.6g}  (dispatches {r[3]})")
    else:
        for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            print(f"  {r[0][:90]:90s} calls={r[1]} total_us={r[2]/1e3:.1f} avg_us={r[3]/1e3:.1f} {r[4]:.1f}%")
