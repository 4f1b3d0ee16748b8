"""rocprofv3 --kernel-trace --stats output directory -> markdown table of the 40 heaviest kernels (+ the nfa:: share).
usage: python tools/kernel_summary.py <rocprof_out_dir> > table.md"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("| kernel | calls | avg us | total ms | % of GPU time |\n|---|---|---|---|---|")
for r in rows[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("| `%s` | %s | %.2f | %.2f | %.1f |" % (n[:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6,
                                               100 * float(r["TotalDurationNs"]) / tot))
nfa = sum(float(r["TotalDurationNs"]) for r in rows if "nfa::" in r["Name"])
print("\nall nfa:: kernels: %.2f ms of %.2f ms GPU time (%.1f %%)" % (nfa / 1e6, tot / 1e6, 100 * nfa / tot))
