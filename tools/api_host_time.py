"""Host time of the two API calls of a training step, closures excluded (time.perf_counter around the calls and around the
user closures; the GPU is NOT synchronised: this is what the interpreter + launches + the two read-backs cost the host)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as nerfacc

dev = torch.device("cuda:0")
torch.manual_seed(42)
field = bench.DenseGridField(bench.AABB, 128).to(dev)
est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
est.train()
for _ in range(4):
    est._update(step=0, occ_eval_fn=lambda x: field.query_density(x) * bench.RENDER_STEP, occ_thre=1e-2)
pool_o, pool_d = bench.make_ray_pool(1 << 20, 42, dev)
bk = torch.ones(3, device=dev)
n = 6500
acc = {"sampling": 0.0, "sigma_fn": 0.0, "rendering": 0.0, "rgb_sigma_fn": 0.0}

def step():
    idx = torch.randint(0, 1 << 20, (n,), device=dev)
    ro, rd = pool_o[idx], pool_d[idx]

    def sigma_fn(t0, t1, ri):
        a = time.perf_counter()
        pos = ro[ri] + rd[ri] * (t0 + t1)[:, None] / 2.0
        out = field.query_density(pos).squeeze(-1)
        acc["sigma_fn"] += time.perf_counter() - a
        return out

    def rgb_sigma_fn(t0, t1, ri):
        a = time.perf_counter()
        pos = ro[ri] + rd[ri] * (t0 + t1)[:, None] / 2.0
        rgb, sig = field(pos)
        out = rgb, sig.squeeze(-1)
        acc["rgb_sigma_fn"] += time.perf_counter() - a
        return out

    a = time.perf_counter()
    ri, t0, t1 = est.sampling(ro, rd, sigma_fn=sigma_fn, near_plane=0.0, far_plane=1e10, render_step_size=bench.RENDER_STEP,
                              stratified=True, cone_angle=0.0, alpha_thre=0.0)
    b = time.perf_counter()
    rgb, op, dp, _ = nerfacc.rendering(t0, t1, ri, n_rays=n, rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bk)
    c = time.perf_counter()
    acc["sampling"] += b - a
    acc["rendering"] += c - b
    rgb.sum().backward()

for _ in range(30):
    step()
torch.cuda.synchronize()
for k in acc:
    acc[k] = 0.0
K = 200
for _ in range(K):
    step()
torch.cuda.synchronize()
print("per step, us:  sampling %.1f (of which sigma_fn %.1f)   rendering %.1f (of which rgb_sigma_fn %.1f)" % (
    acc["sampling"] / K * 1e6, acc["sigma_fn"] / K * 1e6, acc["rendering"] / K * 1e6, acc["rgb_sigma_fn"] / K * 1e6))
print("this package's own host time: sampling %.1f us, rendering %.1f us" % ((acc["sampling"] - acc["sigma_fn"]) / K * 1e6,
                                                                             (acc["rendering"] - acc["rgb_sigma_fn"]) / K * 1e6))
