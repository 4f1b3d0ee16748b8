"""Per streaming kernel: where its waves' cycles go (tools/pmc_streaming_units.sh).  parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waiting at
s_waitcnt: memory), stalled = SQ_WAIT_INST_ANY (issue stall), active = SQ_ACTIVE_INST_ANY; VALU share of the launch = quad-cycles of VALU
issue over 1024 SIMDs x the kernel's duration; waves in flight = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES per SE-normalised unit."""
import collections, csv, glob, os, sys
out_dir, logn = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out_dir, "units", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out_dir, "units", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
KEEP = ("weight_fwd_kernel", "weight_bwd_kernel", "rendering_fwd_kernel", "rendering_bwd_kernel", "accumulate_kernel<3", "accumulate_kernel<1",
        "scan_keyed_kernel", "scan_packed_kernel", "visibility_mask_kernel", "visibility_compact_kernel", "k_copy")
lines = [f"N = 2^{logn}.  One rocprofv3 pass (SQ counters only, with --kernel-trace).  Times under the profiler.", "",
         "| kernel | us | waves | parked at s_waitcnt | issue-stalled | issuing | VALU instructions per wave | VALU busy (of 1024 SIMDs x time) | waves resident per SIMD |",
         "|---|---|---|---|---|---|---|---|---|"]
for sub in KEEP:
    ks = [k for k in agg if sub in k]
    if not ks:
        continue
    m = lambda c: sum(sum(agg[k][c]) / max(len(agg[k][c]), 1) for k in ks) / len(ks)
    us = sum(sum(dur[k]) / max(len(dur[k]), 1) for k in ks if k in dur) / max(sum(1 for k in ks if k in dur), 1)
    wc, wa, wi, ac, av, iv, nw = m("SQ_WAVE_CYCLES"), m("SQ_WAIT_ANY"), m("SQ_WAIT_INST_ANY"), m("SQ_ACTIVE_INST_ANY"), m("SQ_ACTIVE_INST_VALU"), m("SQ_INSTS_VALU"), m("SQ_WAVES")
    cyc = us * 1e-6 * 2.1e9                                   # shader cycles of the launch at the loaded clock (tools/ubench/clock_probe.hip)
    lines.append(f"| `{sub}` | {us:.1f} | {nw:.0f} | {100 * wa / wc:.0f} % | {100 * wi / wc:.0f} % | {100 * ac / wc:.0f} % | {iv / max(nw, 1):.0f} | "
                 f"{100 * 4 * av / (1024 * cyc):.0f} % | {4 * wc / (1024 * cyc):.1f} |")
open(os.path.join(out_dir, "units_table.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
