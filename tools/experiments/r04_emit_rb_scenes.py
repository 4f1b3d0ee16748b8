"""round 4: rays per wave of the tile emit form (`emit_rb`) on the scenes of tools/scenes.py; emit time in us (HIP events)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import nerfacc_amd, scenes
from nerfacc_amd import cuda as C
from scene_sweep import time_call, STEP
dev = "cuda:0"
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("lego", "shell", "drums", "ship", "noise", "ficus")):
    occ = torch.from_numpy(scenes.occupancy_grid(name, res)).to(dev)
    aabb = torch.from_numpy(scenes.AABB[None].copy()).to(dev)
    for n in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        o, d = (torch.from_numpy(x).to(dev) for x in scenes.rays(n, seed=11))
        near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
        call = lambda: C.sample_occgrid(o, d, occ, aabb, near, far, STEP, 0.0)
        t = {}
        for rb in range(7):
            with nerfacc_amd.options(emit_rb=rb, emit="tiles"):
                _, _, t[rb] = time_call(call, 10)
        with nerfacc_amd.options(emit="samples"):
            _, _, ts = time_call(call, 10)
        out, _, ta = time_call(call, 10)             # the automatic choice (round 5: the kernel halves the host's block by runs per ray)
        runs_per_ray = (int(out[3][:, 1].gt(0).sum()), 0)[1]
        print(f"{res}^3 {name:6s} {n:6d} rays  " + "  ".join(f"rb{k} {v:6.1f}" for k, v in t.items()) + f"   samples-form {ts:6.1f}   auto {ta:6.1f} ({ta / min(t.values()):.2f}x best rb)", flush=True)
