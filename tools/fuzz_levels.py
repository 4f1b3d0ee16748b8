"""Long differential fuzz of the fused sampling call on MULTI-LEVEL grids: random level counts (2..8), resolutions, occupancy
kinds, rays from inside / outside / axis-aligned, near / far planes, cone_angle 0 (and > 0 with --cone); the
segment-per-lane count pass (NFA_SEGMENTS=1) and the lane-per-ray one (NFA_SEGMENTS=0) vs the CPU oracle, bit for bit
(ray_indices, t_starts, t_ends, packed_info, terminate planes).  Generators and checkers: tests/fuzz_cases.py.

    python tools/fuzz_levels.py [n_cases] [seed] [--cone]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases as F

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
cones = (0.0, 0.004, 0.02, 0.1) if "--cone" in sys.argv else (0.0,)
n_cases = int(argv[0]) if len(argv) > 0 else 60
seed0 = int(argv[1]) if len(argv) > 1 else 0
bad = total = 0
t_begin = time.time()
for case in range(n_cases):
    c = F.fused_levels_case(np.random.default_rng(seed0 * 1000003 + case), cones=cones)
    if c["cone"] == 0.0:
        b, n = F.check_fused(c, "NFA_SEGMENTS", F.SEGMENT_FORMS)
        b2, _ = F.check_fused(c, "NFA_SEG_P", F.SEG_P_FORMS)
        b += b2
    else:
        b, n = F.check_fused(c, "NFA_CONE", F.CONE_FORMS)
        b2, _ = F.check_fused(c, "NFA_CONE_P", F.CONE_P_FORMS)
        b += b2
    bad += len(b); total += n
    for line in b:
        print("MISMATCH", f"case {case}", line, flush=True)
print(f"{n_cases} cases x 2 count passes ({total} oracle samples in total), {bad} mismatches, {time.time() - t_begin:.0f} s")
sys.exit(1 if bad else 0)
