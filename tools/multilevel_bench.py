"""Unbounded-scene configuration of the reference (train_ngp_nerf_occ.py, mip-NeRF 360 branch: 4 grid
levels, cone_angle 0.004, near 0.2, step 1e-3-ish): sampling traversal on the GPU vs the CPU oracle.

    python tools/multilevel_bench.py [n_rays]
"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from nerfacc_amd import cuda as C

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = np.random.default_rng(0)
res, levels = 128, 4
c = (np.arange(res) + 0.5) / res * 2 - 1
X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
occ = []
for l in range(levels):                                   # a blob at the centre + sparse far clutter
    s = 2.0**l
    blob = (X * s) ** 2 + (Y * s) ** 2 + (Z * s) ** 2 < 0.5**2
    clutter = g.random((res, res, res)) < (float(os.environ.get("ML_CLUTTER", "0.002")) if l else 0.0)
    occ.append(blob | clutter)
occ = np.stack(occ)
aabbs = np.stack([np.array([-1, -1, -1, 1, 1, 1], np.float32) * 2.0**l for l in range(levels)])
v = g.normal(size=(R, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
o = (0.6 * v).astype(np.float32)                          # cameras inside the first level, looking around
d = g.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d = d.astype(np.float32)
near, far = np.full(R, 0.2, np.float32), np.full(R, 1e10, np.float32)
O, D, OCC, AABB, NEAR, FAR = T(o), T(d), T(occ), T(aabbs), T(near), T(far)

def gpu_ms(fn, reps=10):
    for _ in range(3): fn()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]

cases = ((1e-3, 0.004), (1e-3, 0.0), (4e-3, 0.0))
if os.environ.get("ML_ONLY_LATTICE"):
    cases = ((1e-3, 0.0),)
if os.environ.get("ML_ONLY_CONE"):
    cases = ((1e-3, 0.004),)
for step, cone in cases:
    ri, ts, te, pk = C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, step, cone)
    t0 = time.perf_counter()
    iv, sm, _ = oracle.traverse_grids(o, d, occ, aabbs, near, far, step, cone)
    cpu = (time.perf_counter() - t0) * 1e3
    assert os.environ.get("ML_NO_CHECK") or np.array_equal(ri.cpu().numpy(), sm["ray_indices"]) and np.array_equal(ts.cpu().numpy(), iv["vals"][iv["is_left"]])
    ms = gpu_ms(lambda: C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, step, cone))
    print(f"{levels} levels {res}^3, {R} rays, step {step:g}, cone {cone:g}: {ri.shape[0]} samples  GPU {ms*1e3:9.1f} us   CPU oracle {cpu:8.2f} ms   x{cpu/ms:6.0f}")
