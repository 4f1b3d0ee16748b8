"""A/B of the two host faces (torch extension vs ctypes) on the pieces of a training step: wall time per piece,
GPU drained after every repetition (latency) and only at the end (throughput)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import nerfacc_amd as nerfacc
from nerfacc_amd.cuda import _backend as B
dev = torch.device("cuda:0")
torch.manual_seed(0)
field = bench.DenseGridField(bench.AABB, 128).to(dev)
est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
est.train()
for _ in range(4):
    est._update(step=0, occ_eval_fn=lambda x: field.query_density(x) * bench.RENDER_STEP, occ_thre=1e-2)
pool_o, pool_d = bench.make_ray_pool(1 << 18, 42, dev)
bk = torch.ones(3, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
O, D = pool_o[:n].contiguous(), pool_d[:n].contiguous()
opt = torch.optim.Adam(field.parameters(), lr=1e-3, fused=True)
def sampling_only():
    return est.sampling(O, D, render_step_size=bench.RENDER_STEP, stratified=True)
def sampling_sigma():
    def sigma_fn(ts, te, ri):
        return field.query_density(O[ri] + D[ri] * (ts + te)[:, None] / 2.0).squeeze(-1)
    return est.sampling(O, D, sigma_fn=sigma_fn, render_step_size=bench.RENDER_STEP, stratified=True)
def full_step():
    rgb, _, _, ns = bench.render_rays_reference_style(field, est, O, D, bk, True)
    opt.zero_grad()
    (rgb.square().mean() * 1024).backward()
    opt.step()
def run(fn, reps, sync_each):
    for _ in range(15): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
faces = {"ext": (B._hip if hasattr(B, "_hip") else B._C, "ext"), "ctypes": (B._CtypesC, "ctypes")}
ext_mod = B._C
for rnd in range(2):
    for name in ("ext", "ctypes", "ext", "ctypes"):
        B._C, B.BACKEND = (ext_mod, "ext") if name == "ext" else (B._CtypesC, "ctypes")
        B.RaySegmentsSpec = B._C.RaySegmentsSpec
        print(f"{name:7s} sampling {run(sampling_only, 200, False):7.1f}  sampling+sigma_fn {run(sampling_sigma, 200, False):7.1f}  "
              f"full step {run(full_step, 200, False):7.1f} us   (drained each rep: {run(full_step, 100, True):7.1f})")
