"""Print a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <dir-or-csv>"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else max(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)  # the newest run
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-52:]
    print("%-52s calls=%5s avg_us=%10.1f total_ms=%9.2f pct=%6.2f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                   float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
