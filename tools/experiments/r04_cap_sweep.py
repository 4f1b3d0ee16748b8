import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import nerfacc_amd, scenes
from nerfacc_amd import cuda as C
from scene_sweep import time_call, STEP
dev = "cuda:0"
for name in ("lego", "shell", "drums", "materials", "ship"):
    occ = torch.from_numpy(scenes.occupancy_grid(name, 256)).to(dev)
    aabb = torch.from_numpy(scenes.AABB[None].copy()).to(dev)
    for n in (10000, 24000, 48000, 96000):
        o, d = (torch.from_numpy(x).to(dev) for x in scenes.rays(n, seed=11))
        near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
        call = lambda: C.sample_occgrid(o, d, occ, aabb, near, far, STEP, 0.0)
        t = {}
        for P in (16, 8):
            for cap in (16, 24, 32):
                with nerfacc_amd.options(split_p=P, split_l2=1, split_cap=cap):
                    _, t[f"P{P}c{cap}"], _ = time_call(call, 8)
        with nerfacc_amd.options(split_p=4):
            _, t["P4"], _ = time_call(call, 8)
        print(f"{name:10s} {n:6d} " + "  ".join(f"{k} {v:6.1f}" for k, v in t.items()), flush=True)
