"""GPU idle gaps of a rocprofv3 --kernel-trace run (csv): busy time, span, and which kernels the
GPU waits *before* (gap attributed to the kernel that ends it).  python tools/gap_analysis.py trace.csv [skip_first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 3      # drop the warm-up third
rows = rows[skip:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = collections.defaultdict(lambda: [0, 0])
prev_end = int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    g = int(r["Start_Timestamp"]) - prev_end
    if g > 0:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        gaps[k][0] += g
        gaps[k][1] += 1
    prev_end = max(prev_end, int(r["End_Timestamp"]))
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %)  idle {(span-busy)/1e6:.2f} ms")
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  idle before {k:72s} {g/1e6:8.2f} ms  in {n:5d} gaps  avg {g/n/1e3:7.1f} us")
