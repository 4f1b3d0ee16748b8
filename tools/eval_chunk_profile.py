"""Where does an 8192-ray eval chunk of the reference's render_image_with_occgrid loop spend its ~200 us?  perf_counter around the
pieces of one chunk (GPU not synchronised: host time), averaged over the 79 chunks of 800x800 frames, and the GPU-side kernel time.
    python tools/eval_chunk_profile.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
import nerfacc_amd as nerfacc
from ref_examples_check import frame_rays
import collections
Rays = collections.namedtuple("Rays", ("origins", "viewdirs"))
dev = torch.device("cuda:0")
torch.manual_seed(42)
field = bench.DenseGridField(bench.AABB, 128).to(dev).eval()
est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
est.train()
for _ in range(4):
    est._update(step=0, occ_eval_fn=lambda x: field.query_density(x) * bench.RENDER_STEP, occ_thre=1e-2)
est.eval()
rays = frame_rays(Rays, 0, 800, dev)
O, D = rays.origins.reshape(-1, 3), rays.viewdirs.reshape(-1, 3)
bk = torch.ones(3, device=dev)
T = collections.defaultdict(float)
def frame():
    for i in range(0, O.shape[0], 8192):
        o, d = O[i:i + 8192], D[i:i + 8192]
        def sigma_fn(ts, te, ri):
            a = time.perf_counter()
            if ts.shape[0] == 0:
                return torch.empty((0,), device=dev)
            pos = o[ri] + d[ri] * (ts + te)[:, None] / 2.0
            s = field.query_density(pos).squeeze(-1)
            T["sigma_fn"] += time.perf_counter() - a
            return s
        def rgb_sigma_fn(ts, te, ri):
            a = time.perf_counter()
            if ts.shape[0] == 0:
                return torch.empty((0, 3), device=dev), torch.empty((0,), device=dev)
            pos = o[ri] + d[ri] * (ts + te)[:, None] / 2.0
            rgb, s = field(pos, d[ri])
            T["rgb_sigma_fn"] += time.perf_counter() - a
            return rgb, s.squeeze(-1)
        a = time.perf_counter()
        ri, ts, te = est.sampling(o, d, sigma_fn=sigma_fn, render_step_size=bench.RENDER_STEP, stratified=False)
        b = time.perf_counter()
        nerfacc.rendering(ts, te, ri, n_rays=o.shape[0], rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bk)
        c = time.perf_counter()
        T["sampling"] += b - a; T["rendering"] += c - b
with torch.no_grad():
    frame(); torch.cuda.synchronize(); T.clear()
    t0 = time.perf_counter()
    for _ in range(3): frame()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3
    n_chunks = -(-O.shape[0] // 8192)
    print(f"frame {wall*1e3:.2f} ms, {n_chunks} chunks, {wall/n_chunks*1e6:.0f} us per chunk; host per chunk, us: " +
          "  ".join(f"{k} {v/3/n_chunks*1e6:.0f}" for k, v in T.items()))
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        frame(); torch.cuda.synchronize()
    tot = sum(float(getattr(e, "device_time", 0) or 0) for e in prof.events() if "cuda" in str(getattr(e, "device_type", "")).lower())
    nfa = sum(float(getattr(e, "device_time", 0) or 0) for e in prof.events() if "nfa::" in e.name)
    print(f"GPU kernel time per chunk {tot/n_chunks:.0f} us, of which nfa:: {nfa/n_chunks:.0f} us")
