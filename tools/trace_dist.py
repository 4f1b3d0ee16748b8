import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = [int(r["Grid_Size_X"]) for r in rows]
n = len(d)
for a, b in ((0, 100), (100, 500), (500, 1000), (1000, 1500), (1500, n)):
    seg = sorted(d[a:b]); gs = g[a:b]
    if seg: print(f"calls {a}-{b}: median {seg[len(seg)//2]:.1f} p90 {seg[int(len(seg)*0.9)]:.1f} max {seg[-1]:.1f} grid median {sorted(gs)[len(gs)//2]}")
