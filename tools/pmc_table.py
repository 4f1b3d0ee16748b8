"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files -> markdown.
usage: python tools/pmc_table.py out.md file1.csv [file2.csv ...]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for f in sys.argv[2:]:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-64:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
names = sorted({c for v in agg.values() for c in v})
lines = ["| kernel | " + " | ".join(names) + " |", "|---|" + "---|" * len(names)]
for k in sorted(agg):
    if "nfa" not in k:
        continue
    lines.append(f"| `{k}` | " + " | ".join(f"{agg[k][n_] / max(cnt[k][n_], 1):.4g}" if n_ in agg[k] else "" for n_ in names) + " |")
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
