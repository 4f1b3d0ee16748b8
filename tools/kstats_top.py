"""print the top kernels of a rocprofv3 --stats kernel_stats.csv:  python tools/kstats_top.py file.csv [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "")
    print("%-100s calls=%6s avg_us=%9.1f total_ms=%8.2f" % (name[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
