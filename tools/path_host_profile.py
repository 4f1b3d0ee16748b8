"""Where the host time of the path-only step goes (bench.py: path_only_loop): cProfile of 300 steps of estimator.sampling +
nerfacc.rendering + backward with a free field, top functions by cumulative time."""
import cProfile, os, pstats, sys, time
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
free_sig = torch.rand(1 << 21, device=dev) * 20.0
free_rgb = torch.rand(1 << 21, 3, device=dev)
T = {"draw": 0.0, "sampling": 0.0, "leaves": 0.0, "rendering": 0.0, "backward": 0.0}

def step():
    a = time.perf_counter()
    idx = torch.randint(0, 1 << 20, (n,), device=dev)
    ro, rd = pool_o[idx], pool_d[idx]
    b = time.perf_counter()
    ri, t0, t1 = est.sampling(ro, rd, sigma_fn=lambda x, y, r: free_sig[:x.shape[0]], near_plane=0.0, far_plane=1e10,
                              render_step_size=bench.RENDER_STEP, stratified=True, cone_angle=0.0, alpha_thre=0.0)
    c = time.perf_counter()
    k = t0.shape[0]
    leaves = (free_rgb[:k].detach().requires_grad_(True), free_sig[:k].detach().requires_grad_(True))
    d = time.perf_counter()
    rgb, _, _, _ = nerfacc.rendering(t0, t1, ri, n_rays=n, rgb_sigma_fn=lambda x, y, r: leaves, render_bkgd=bk)
    e = time.perf_counter()
    rgb.sum().backward()
    f = time.perf_counter()
    for key, v in zip(T, (b - a, c - b, d - c, e - d, f - e)):
        T[key] += v

for _ in range(50):
    step()
torch.cuda.synchronize()
for key in T:
    T[key] = 0.0
K = 300
t0_ = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
wall = time.perf_counter() - t0_
print("wall per step %.1f us; host per step, us: " % (wall / K * 1e6) + "  ".join(f"{k} {v / K * 1e6:.1f}" for k, v in T.items()))
if "--cprofile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(K):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
