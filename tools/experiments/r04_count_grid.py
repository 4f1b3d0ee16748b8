"""round 4: lanes per ray of the one-level count pass against the ray count, on several scenes and two grid sizes
(tools/scenes.py) — the data behind count_lanes_per_ray (grid.hip).  Count-pass time in us from the call's HIP events.
    python tools/experiments/r04_count_grid.py [scene ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import nerfacc_amd, scenes
from nerfacc_amd import cuda as C
from scene_sweep import time_call, STEP

FORMS = {"P16": dict(split_p=16, split_l2=1), "P8": dict(split_p=8), "P4": dict(split_p=4), "P2": dict(split_p=2), "P1 l2": dict(split_p=1, count_l2=1),
         "P1 lds": dict(split_p=1, count_l2=0)}
# (grids read from L2: `split_cap` = 16 | 32 forces the boundary-list capacity of the 8- and 16-lane forms; the second table of
# profiles/r04_count_pass.md was taken with the automatic capacity — 16 entries on every scene but `noise`)
names = [a for a in sys.argv[1:] if not a.startswith("-")] or ["lego", "ficus", "ship", "speck", "noise", "drums"]
dev = "cuda:0"
for res in (128, 256):
    for name in names:
        occ = torch.from_numpy(scenes.occupancy_grid(name, res)).to(dev)
        aabb = torch.from_numpy(scenes.AABB[None].copy()).to(dev)
        for n in (6000, 12000, 24000, 48000, 96000, 192000):
            o, d = (torch.from_numpy(x).to(dev) for x in scenes.rays(n, seed=11))
            near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
            call = lambda: C.sample_occgrid(o, d, occ, aabb, near, far, STEP, 0.0)
            nerfacc_amd.reset_options()
            _, c_auto, _ = time_call(call, 6)
            t = {}
            for tag, f in FORMS.items():
                with nerfacc_amd.options(**f):
                    _, t[tag], _ = time_call(call, 6)
            best = min(t, key=t.get)
            print(f"{res:3d}^3 {name:6s} {n:7d} rays  auto {c_auto:7.1f}  " + "  ".join(f"{k} {v:7.1f}" for k, v in t.items()) + f"   best {best} ({c_auto / t[best]:.2f}x)", flush=True)
