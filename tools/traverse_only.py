"""Micro-driver: run the two traversal passes N times on the bench's ray batch (for rocprofv3)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as nerfacc
from nerfacc_amd import cuda as C

dev = torch.device("cuda:0")
torch.manual_seed(42)
field = bench.DenseGridField(bench.AABB, 128).to(dev)
est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
est.train()
for _ in range(4):
    est._update(step=0, occ_eval_fn=lambda x: field.query_density(x) * bench.RENDER_STEP, occ_thre=1e-2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13120
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pool_o, pool_d = bench.make_ray_pool(n, 42, dev)
near = torch.rand(n, device=dev) * bench.RENDER_STEP
far = torch.full((n,), 1e10, device=dev)
for _ in range(iters):
    ri, ts, te, pk = C.sample_occgrid(pool_o, pool_d, est.binaries, est.aabbs, near, far, bench.RENDER_STEP, 0.0)
torch.cuda.synchronize()
print("rays", n, "samples", ri.shape[0], "occupied bricks", int(C.packed_bricks(est.binaries)[32768].item()))
