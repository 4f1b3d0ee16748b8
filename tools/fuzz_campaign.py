"""Long differential fuzz (not part of the suite; tests/test_gpu_fuzz.py runs a time-boxed sample of the same generators on
every `-m gpu` run): random single-level grids through the fused sampling call under every lanes-per-ray setting, then the
reference-API call (traverse_grids) on multi-level grids with cone angles, per-voxel mode, step limits + over-allocation and
ray masks — all vs the CPU oracle, bit for bit.  Generators and checkers: tests/fuzz_cases.py.

    python tools/fuzz_campaign.py [n_cases] [seed]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases as F

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = total = 0
t_begin = time.time()
for case in range(n_cases):
    b, n = F.check_fused(F.fused_single_case(np.random.default_rng(seed0 * 100003 + case)), "NFA_SPLIT_P", F.SPLIT_P_FORMS)
    bad += len(b); total += n
    for line in b:
        print("MISMATCH", f"case {case}", line, flush=True)
    b, n = F.check_fused(F.fused_single_case(np.random.default_rng(seed0 * 100003 + case + 31)), None, F.IMAGE_FORMS)
    bad += len(b); total += n
    for line in b:
        print("MISMATCH (image)", f"case {case}", line, flush=True)
    b, n = F.check_fused(F.fused_single_case(np.random.default_rng(seed0 * 100003 + case + 53), cones=(0.0, 0.0, 0.004, 0.3)), "NFA_EMIT", F.EMIT_FORMS)
    bad += len(b); total += n
    for line in b:
        print("MISMATCH (emit)", f"case {case}", line, flush=True)
    # round 6: ray counts inside the window of the single-launch sampling call, fused (twice) / in three launches
    b, n = F.check_fused(F.fused_single_case(np.random.default_rng(seed0 * 100003 + case + 97), ray_counts=(3072, 3105, 4097, 6564, 8191, 8192)), "NFA_FUSED_SAMPLE", F.FUSED_FORMS)
    bad += len(b); total += n
    for line in b:
        print("MISMATCH (fused)", f"case {case}", line, flush=True)
    if case % 3 == 0:      # one level with a cone angle: the two-phase kernel with a lane per ray vs the general kernel
        b, n = F.check_fused(F.fused_single_case(np.random.default_rng(seed0 * 100003 + case + 77), cones=(0.004, 0.05, 0.3)), "NFA_CONE", F.CONE_FORMS)
        bad += len(b); total += n
        for line in b:
            print("MISMATCH (cone)", f"case {case}", line, flush=True)
print(f"{n_cases} cases x ({len(F.SPLIT_P_FORMS)} lane settings + {len(F.IMAGE_FORMS)} image placements / list capacities + {len(F.EMIT_FORMS)} emit forms) ({total} oracle samples in total), {bad} mismatches, {time.time() - t_begin:.0f} s")
bad2 = tot2 = 0
t_begin = time.time()
for case in range(n_cases // 4):
    b, n = F.check_api(F.api_case(np.random.default_rng(seed0 * 7919 + 5000 + case)))
    bad2 += len(b); tot2 += n
    for line in b:
        print("MISMATCH (API)", f"case {case}", line, flush=True)
print(f"traverse_grids API: {n_cases // 4} cases ({tot2} oracle samples), {bad2} mismatches, {time.time() - t_begin:.0f} s")
sys.exit(1 if bad + bad2 else 0)
