"""FETCH_SIZE / WRITE_SIZE per kernel from tools/pmc_streaming.sh, calibrated on the tuned copy of known size that runs in the
same passes (guide: FETCH_SIZE counts half of a wide coalesced read on gfx950, WRITE_SIZE is uncalibrated)."""
import collections, csv, glob, os, sys
out_dir, logn = sys.argv[1], int(sys.argv[2])
log = open(os.path.join(out_dir, "FETCH_SIZE.log")).read()
N = int(log.split("N = ")[1].split()[0]); R = int(log.split("R = ")[1].split()[0])
alg = {   # kernel substring -> (read bytes, written bytes) per call, DESIGN.md section 3
    "k_copy": (16 * N, 16 * N),
    "weight_fwd_kernel": (20 * N, 12 * N), "weight_bwd_kernel": (32 * N, 4 * N),
    "rendering_fwd_kernel": (32 * N, 12 * N + 20 * R), "rendering_bwd_kernel": (44 * N + 20 * R, 16 * N),
    "accumulate_kernel<3": (24 * N, 12 * R), "accumulate_kernel<1": (12 * N, 4 * R),
    "scan_keyed_kernel": (12 * N, 4 * N), "scan_packed_kernel": (4 * N + 16 * R, 4 * N),
    "visibility_mask_kernel": (20 * N, N), "visibility_compact_kernel": (17 * N, 16 * N),
}
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(out_dir, c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c:
                continue
            k = row["Kernel_Name"]
            agg[k] += float(row["Counter_Value"]); cnt[k] += 1
    val[c] = {k: agg[k] / cnt[k] for k in agg}
def find(c, sub):
    xs = [v for k, v in val[c].items() if sub in k]
    return sum(xs) / len(xs) if xs else float("nan")
cal_r = 16 * N / find("FETCH_SIZE", "k_copy")
cal_w = 16 * N / find("WRITE_SIZE", "k_copy")
lines = [f"N = {N} samples, R = {R} rays.  Calibration on the tuned 16-byte-lane copy of 16 N bytes each way in the same passes: "
         f"1 FETCH_SIZE unit = {cal_r:.1f} B, 1 WRITE_SIZE unit = {cal_w:.1f} B.", "",
         "| kernel | algorithmic read MB | fetched MB | x | algorithmic write MB | written MB | x |", "|---|---|---|---|---|---|---|"]
for sub, (rb, wb) in alg.items():
    fr, fw = find("FETCH_SIZE", sub) * cal_r, find("WRITE_SIZE", sub) * cal_w
    lines.append(f"| `{sub}` | {rb / 1e6:.1f} | {fr / 1e6:.1f} | {fr / rb:.2f} | {wb / 1e6:.1f} | {fw / 1e6:.1f} | {fw / max(wb, 1):.2f} |")
open(os.path.join(out_dir, "table.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
