"""Diagnostic: host-side cost of one bench-like training step (torch.profiler, CPU activity only):
how many ATen ops / launches the step issues and where the Python/dispatch time goes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as nerfacc
import torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(42)
field = bench.DenseGridField(bench.AABB, 128).to(dev)
est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
occ_fn = lambda x: field.query_density(x) * bench.RENDER_STEP
est.train()
for _ in range(4):
    est._update(step=0, occ_eval_fn=occ_fn, occ_thre=1e-2)
pool_o, pool_d = bench.make_ray_pool(1 << 20, 42, dev)
pool_rgb = torch.rand(1 << 20, 3, device=dev)
bk = torch.ones(3, device=dev)
opt = torch.optim.Adam(field.parameters(), lr=1e-2, eps=1e-15, fused=True)
n = 13120

def step():
    idx = torch.randint(0, 1 << 20, (n,), device=dev)
    ro, rd, pix = pool_o[idx], pool_d[idx], pool_rgb[idx]
    rgb, acc, depth, ns = bench.render_rays(field, est, ro, rd, bk, True)
    opt.zero_grad()
    loss = F.smooth_l1_loss(rgb, pix)
    (loss * 1024).backward()
    opt.step()
    return ns

for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step()
torch.cuda.synchronize()
print("ms/step (no grid update, no collectives):", (time.perf_counter() - t0) * 10)
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(20):
        step()
    torch.cuda.synchronize()
ka = prof.key_averages()
print("ops per step:", sum(e.count for e in ka) / 20)
print(ka.table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=60))
