"""Timeline of the kernels behind the last sketch launch of a rocprofv3 --kernel-trace run (the pair phase + MST of the last step):
tools/pair_timeline.py <dir-or-kernel_trace.csv> [until_us]"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else max(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
until = float(sys.argv[2]) if len(sys.argv) > 2 else 1600.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if "sketch_" in r["Kernel_Name"] and "_kernel" in r["Kernel_Name"]][-1]
t0 = prev = int(rows[last]["End_Timestamp"])
for r in rows[last + 1:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = ("rocprim:" + (n.split("wrapped_")[1][:28] if "wrapped_" in n else n.split("detail::")[-1][:40])) if "rocprim" in n else n.split("(")[0][:50]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t0) / 1000 > until:
        break
    print(f"{(s - t0) / 1000:9.1f} gap={(s - prev) / 1000:7.1f} dur={(e - s) / 1000:8.1f}  {n}")
    prev = e
