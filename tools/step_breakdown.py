"""Diagnostic: where does a bench step spend its wall time?  (sync after every phase)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as nerfacc
import torch.nn.functional as F

dev = torch.device("cuda:0")
torch.manual_seed(42)
field = bench.DenseGridField(bench.AABB, 128).to(dev)
est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
occ_fn = lambda x: field.query_density(x) * bench.RENDER_STEP
est.train()
for _ in range(4):
    est._update(step=0, occ_eval_fn=occ_fn, occ_thre=1e-2)
pool_o, pool_d = bench.make_ray_pool(1 << 20, 42, dev)
bk = torch.ones(3, device=dev)
opt = torch.optim.Adam(field.parameters(), lr=1e-2, eps=1e-15)
n = 13120
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
for it in range(120):
    if it == 20: T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = torch.randint(0, 1 << 20, (n,), device=dev)
    ro, rd = pool_o[idx], pool_d[idx]
    pix = torch.rand(n, 3, device=dev)
    t0 = tick("index", t0)
    def sigma_fn(ts, te, ri):
        pos = ro[ri] + rd[ri] * ((ts + te)[:, None] / 2.0)
        return field.query_density(pos).squeeze(-1)
    near = torch.zeros(n, device=dev); far = torch.full((n,), 1e10, device=dev)
    from nerfacc_amd import cuda as C
    ri, ts, te, pk = C.sample_occgrid(ro, rd, est.binaries, est.aabbs, near, far, 5e-3, 0.0)
    t0 = tick("traverse", t0)
    with torch.no_grad():
        sig = sigma_fn(ts, te, ri)
    t0 = tick("sigma_fn", t0)
    ri, ts, te, _ = C.visibility_compact(ri, ts, te, sig, False, 1e-4, 0.0)
    t0 = tick("visibility", t0)
    def rgb_sigma_fn(ts, te, ri):
        pos = ro[ri] + rd[ri] * ((ts + te)[:, None] / 2.0)
        rgb, s = field(pos)
        return rgb, s.squeeze(-1)
    rgb, opa, dep, ex = nerfacc.rendering(ts, te, ri, n, rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bk)
    t0 = tick("rendering_fwd(+field)", t0)
    loss = F.smooth_l1_loss(rgb, pix)
    opt.zero_grad()
    (loss * 1024).backward()
    t0 = tick("backward(+field)", t0)
    opt.step()
    t0 = tick("adam", t0)
tot = sum(T.values())
for k, v in T.items():
    print(f"{k:28s} {v/100*1e3:8.3f} ms")
print("total", tot / 100 * 1e3, "ms/step ;", ts.shape[0], "samples")
