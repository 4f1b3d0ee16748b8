"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel stats table.
usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for nme, c, s, a, mn, mx in rows:
        short = nme if len(nme) < 110 else nme[:107] + "..."
        lines.append(f"| `{short}` | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/total:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
