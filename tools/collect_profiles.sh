#!/bin/bash
# One pass over everything profiles/ quotes (run on the GPU box from the repo root):  tools/collect_profiles.sh <tag>
# writes gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=${1:-r02}; OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
python bench.py --dump-sampling-state $OUT/${TAG}_sampling_state.npz > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-profile --no-other-mode > $OUT/${TAG}_kt_line.json 2> $OUT/${TAG}_kt.err
python - <<PY > $OUT/${TAG}_bench_kernels.md
import csv, glob
f = glob.glob("$OUT/${TAG}_kt/*kernel_stats.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("| kernel | calls | avg us | total ms | % of GPU time |\n|---|---|---|---|---|")
for r in rows[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"| \`{n[:110]}\` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.2f} | {100*float(r['TotalDurationNs'])/tot:.1f} |")
nfa = sum(float(r["TotalDurationNs"]) for r in rows if "nfa::" in r["Name"])
print(f"\nall nfa:: kernels: {nfa/1e6:.2f} ms of {tot/1e6:.2f} ms GPU time ({100*nfa/tot:.1f} %)")
PY
rm -rf $OUT/${TAG}_kt/*kernel_trace.csv $OUT/${TAG}_kt/*.db
tools/pmc_traverse.sh $OUT/${TAG}_sampling_state.npz $OUT/${TAG}_pmc 20 > /dev/null 2>&1
rm -rf $OUT/${TAG}_pmc/*/*kernel_trace.csv $OUT/${TAG}_pmc/*/*.db
python tools/roofline_sweep.py 24 $OUT/${TAG}_stream24.md > /dev/null 2>&1
python tools/roofline_sweep.py 26 $OUT/${TAG}_stream26.md > /dev/null 2>&1
python tools/backend_overhead.py > $OUT/${TAG}_host_overhead.txt 2>/dev/null
python tools/microbench.py > $OUT/${TAG}_microbench.txt 2>/dev/null
ls -la $OUT
