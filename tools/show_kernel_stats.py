"""Print a rocprofv3 --kernel-trace --stats kernel_stats.csv compactly: python tools/show_kernel_stats.py FILE [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"][5:] if r["Name"].startswith("void ") else r["Name"]
    print("%-72s calls %6s avg %8.1f us %5.1f%%  min %7.1f max %7.1f" % (name[:72], r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["Percentage"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
print("%.3f ms total kernel time" % (tot / 1e6))
