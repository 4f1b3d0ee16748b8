"""The path-only training step (bench.py: path_only_loop — estimator.sampling with the visibility filter, nerfacc.rendering forward and
backward, the field replaced by slices of constant tensors) on the bench's recorded steady state (profiles/r02_sampling_state.npz:
128^3 grid, 6 564 rays), with round 6's single-launch forms switched on and off one at a time: wall microseconds per step.
    python tools/path_ab.py [steps] [--forms=000,100,010,001,111]      digits: fused_sample fused_vis fold_fill (the defaults are 101)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import nerfacc_amd as nerfacc

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(args[0]) if args else 400
forms = next((a.split("=")[1] for a in sys.argv if a.startswith("--forms=")), "000,100,010,001,101,111").split(",")
dev = torch.device("cuda:0")
st = np.load(os.path.join(ROOT, "profiles", "r02_sampling_state.npz"))
res = tuple(int(x) for x in st["res"])
binaries = torch.from_numpy(np.unpackbits(st["binaries_bits"])[: int(np.prod(res))].astype(bool).reshape(res)).to(dev)
est = nerfacc.OccGridEstimator(roi_aabb=st["aabbs"][0].tolist(), resolution=res[1], levels=1).to(dev)
est.binaries = binaries
O, D = torch.from_numpy(st["rays_o"]).to(dev), torch.from_numpy(st["rays_d"]).to(dev)
step_size = float(st["render_step"])
n = O.shape[0]
free_sig = torch.rand(1 << 21, device=dev) * 20.0
free_rgb = torch.rand(1 << 21, 3, device=dev)
bkgd = torch.ones(3, device=dev)
tot = {"k": 0}

def step():
    ri, ts, te = est.sampling(O, D, sigma_fn=lambda a, b, r: free_sig[:a.shape[0]], near_plane=0.0, far_plane=1e10,
                              render_step_size=step_size, stratified=True, cone_angle=0.0, alpha_thre=0.0)
    k = ts.shape[0]
    leaves = (free_rgb[:k].detach().requires_grad_(True), free_sig[:k].detach().requires_grad_(True))
    rgb, _, _, _ = nerfacc.rendering(ts, te, ri, n_rays=n, rgb_sigma_fn=lambda a, b, r: leaves, render_bkgd=bkgd)
    rgb.sum().backward()
    tot["k"] = k

reps = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--reps=")), 5)
wall = {f: [] for f in forms}
for rep in range(reps):                      # (round-robin over the forms: a drifting host clock hits every form alike)
    for f in forms:
        with nerfacc.options(fused_sample=int(f[0]), fused_vis=int(f[1]), fold_fill=int(f[2])):
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            wall[f].append((time.perf_counter() - t0) / steps * 1e6)
for f in forms:
    w = sorted(wall[f])
    print(f"forms {f} (fused_sample fused_vis fold_fill)  median {w[len(w) // 2]:7.1f}  min {w[0]:7.1f}  max {w[-1]:7.1f} us/step over {reps} x {steps} steps   "
          f"rendered samples {tot['k']}", flush=True)
