"""cone_angle > 0 on ONE level (bench scene's grid, 128^3): the two-phase kernel with a lane per ray (NFA_CONE=1) vs the general kernel.
    python tools/cone_single_level.py [n_rays]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as nerfacc
from nerfacc_amd import cuda as C
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(0)
field = bench.DenseGridField(bench.AABB, 128).to(dev)
est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
est.train()
for _ in range(4):
    est._update(step=0, occ_eval_fn=lambda x: field.query_density(x) * bench.RENDER_STEP, occ_thre=1e-2)
o, d = bench.make_ray_pool(R, 3, dev)
near, far = torch.zeros(R, device=dev), torch.full((R,), 1e10, device=dev)
def ms(fn, reps=10):
    for _ in range(3): fn()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1))
    return sorted(out)[len(out) // 2]
res = {}
for mode in ("1", "0"):
    import nerfacc_amd; nerfacc_amd.set_option("cone", mode)
    f = lambda: C.sample_occgrid(o, d, est.binaries, est.aabbs, near, far, bench.RENDER_STEP, 0.004)
    res[mode] = (ms(f) * 1e3, f())
a, b = res["1"][1], res["0"][1]
assert all(torch.equal(x, y) for x, y in zip(a, b))
print(f"one level 128^3, {R} rays, cone 0.004, {a[0].shape[0]} samples: two-phase {res['1'][0]:.1f} us, general {res['0'][0]:.1f} us")
