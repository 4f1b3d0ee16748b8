"""gpurun_out/<tag>_* (tools/collect_profiles.sh) -> profiles/r02_*: the json line, the per-kernel summary (table replaced under its
header), the PMC json + the three rows of its markdown table, the two streaming tables, the micro-benchmark block, the sampling state.
The prose of the markdown files is maintained by hand.    python tools/publish_profiles.py r02e"""
import json, os, re, shutil, sys
tag = sys.argv[1]
G, P = "gpurun_out/" + tag + "_", "profiles/r02_"
line = open(G + "bench_line.json").read().strip().splitlines()[-1]
json.loads(line)
open(P + "bench_line.json", "w").write(line + "\n")
shutil.copy(G + "sampling_state.npz", P + "sampling_state.npz")
# per-kernel summary
old = open(P + "bench_kernels.md").read()
new = open(G + "bench_kernels.md").read().rstrip()
hdr, rest = old[:old.index("| kernel | calls")], old[old.index("\nReading:"):]
open(P + "bench_kernels.md", "w").write(hdr + new + "\n" + rest)
# PMC
j = json.load(open(G.rstrip("_") + "_pmc/pmc_traverse.json"))
j["state"] = "r02_sampling_state.npz"
json.dump(j, open(P + "pmc_traverse.json", "w"), indent=1)
k = j["kernels"]
def row(key):
    e = k[key]
    return (f"| `{key}` | {e['avg_us']:.1f} | {int(e['SQ_WAVES'])} | {e['SQ_INSTS_VALU']:.3g} | {e['SQ_INSTS_SALU']:.3g} | {e['SQ_INSTS_LDS']:.3g} | "
            f"{e['SQ_INSTS_VMEM_RD']:.3g} / {e['SQ_INSTS_VMEM_WR']:.3g} | {e['insts_per_wave']:.0f} | {e['hbm_bytes']/1e6:.2f} MB | {100*e['issue_frac']:.1f} % |")
md = open(P + "pmc_traverse.md").read().split("\n")
md = [row(l.split("`")[1]) if l.startswith("| `traverse_") else l for l in md]
open(P + "pmc_traverse.md", "w").write("\n".join(md))
# streaming tables
s = open(P + "streaming.md").read()
a, b = s.index("# streaming kernels at N = "), s.index("\n**Aligned wave tiles**")
s = s[:a] + open(G + "stream24.md").read().rstrip() + "\n\n" + open(G + "stream26.md").read().rstrip() + "\n" + s[b:]
open(P + "streaming.md", "w").write(s)
# micro-benchmarks
m = open(P + "microbench.md").read()
a = m.index("```\n") + 4
b = m.index("```", a)
rows = [l for l in open(G + "microbench.txt").read().split("\n") if l.startswith("M")]
open(P + "microbench.md", "w").write(m[:a] + "\n".join(rows) + "\n" + m[b:])
print("published", tag)
