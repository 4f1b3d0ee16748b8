"""rocprofv3 csv files of tools/pmc_traverse.sh -> profiles-style json + markdown (per launch averages of the three
traversal kernels: HBM bytes from FETCH_SIZE / WRITE_SIZE, instructions issued, issue-slot occupancy)."""
import collections, csv, glob, json, os, sys
import numpy as np
state, out = sys.argv[1], sys.argv[2]
st = np.load(state)
R, cand = int(st["rays_o"].shape[0]), int(st["candidates"])
# runs with `--rays=N` (tools/pmc_traverse.sh's 4th argument) tile the state's batch: the header has to say what was launched, not what
# the state holds (r04_pmc_traverse_1m.json said 6564 rays for a 10^6-ray run).  The replay prints both.
for lg in glob.glob(os.path.join(out, "trace.log")):
    for line in open(lg):
        if line.startswith("rays ") and " candidates " in line:
            w = line.split()
            R, cand = int(w[1]), int(w[3])
KERNELS = ("traverse_count", "traverse_offsets", "traverse_emit")
def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "*kernel_trace.csv")):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k:
            dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
def avg(k, c, skip=1):
    v = agg[k].get(c, [])
    v = v[skip:] if len(v) > skip else v
    return float(np.mean(v)) if v else None
res = {"state": os.path.basename(state) + ("" if R == int(st["rays_o"].shape[0]) else f" tiled to {R} rays"), "rays_per_launch": R, "candidate_samples_per_launch": cand, "kernels": {}}
CLOCK_HZ, SIMDS = 2.4e9, 1024
tot_bytes = 0.0
most = max((len(v) for v in dur.values()), default=0)
for k in KERNELS:
    d = dur.get(k, [])
    # round 6: at the training size the call is ONE launch (the fused form of traverse_count_split_kernel: count, offsets by look-back,
    # emit); the offsets / emit kernels then run once, in the replay's first call (no guess of the output size yet), and are not part
    # of a steady-state launch's traffic
    steady = len(d) * 2 >= most
    us = float(np.mean(d[1:])) if len(d) > 1 else (d[0] if d else None)
    fetch, write = avg(k, "FETCH_SIZE"), avg(k, "WRITE_SIZE")
    e = {"avg_us": us, "launches": len(d), "in_every_call": bool(steady), "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write}
    if fetch is not None and write is not None:
        e["hbm_bytes"] = (2.0 * fetch + write) * 1024.0          # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md)
        if steady:
            tot_bytes += e["hbm_bytes"]
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAVES",
              "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_SALU"):
        v = avg(k, c)
        if v is not None:
            e[c] = v
    if us and "SQ_INSTS_VALU" in e:
        # issue-slot model: one wave64 VALU instruction holds its SIMD's VALU for 2 cycles (v_fma_f32 row of the guide's
        # table), a scalar / LDS / memory instruction for 1 issue cycle; 1024 SIMDs x 2.4 GHz
        slots = 2.0 * e["SQ_INSTS_VALU"] + e.get("SQ_INSTS_SALU", 0) + e.get("SQ_INSTS_LDS", 0) + e.get("SQ_INSTS_VMEM_RD", 0) + e.get("SQ_INSTS_VMEM_WR", 0) + e.get("SQ_INSTS_SMEM", 0)
        e["issue_frac"] = slots / (SIMDS * CLOCK_HZ * us * 1e-6)
        e["insts_per_wave"] = (e["SQ_INSTS_VALU"] + e.get("SQ_INSTS_SALU", 0) + e.get("SQ_INSTS_LDS", 0)) / max(e.get("SQ_WAVES", 1), 1)
    res["kernels"][k] = e
res["hbm_bytes_per_launch"] = tot_bytes or None
c = res["kernels"].get("traverse_count", {})
if "issue_frac" in c:
    res["issue"] = {"kernel": "traverse_count_split_kernel" + ("" if res["kernels"].get("traverse_emit", {}).get("in_every_call") else " (fused: count + look-back + emit)"), "frac_of_issue_slots": c["issue_frac"], "valu": c.get("SQ_INSTS_VALU"),
                    "salu": c.get("SQ_INSTS_SALU"), "lds": c.get("SQ_INSTS_LDS"), "waves": c.get("SQ_WAVES"),
                    "model": "(2*VALU + SALU + LDS + VMEM + SMEM wave-instructions) / (1024 SIMDs * 2.4 GHz * kernel time)"}
json.dump(res, open(os.path.join(out, "pmc_traverse.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
