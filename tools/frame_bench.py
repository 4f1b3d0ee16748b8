"""800x800 test frames of bench.py's scene (SURVEY.md 8f-3 / M-eval): chunked training-path
rendering (examples/utils.py:54-167, 8192-ray chunks as the reference evaluates) vs the test-time
iterative marcher (:267-439), reference-API composition vs the fused per-round call.

    python tools/frame_bench.py [n_frames]
"""
import math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import bench
import nerfacc_amd as nerfacc
import utils as U

dev = torch.device("cuda:0")
torch.manual_seed(42)
field = bench.DenseGridField(bench.AABB, 128).to(dev).eval()
est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
est.train()
for _ in range(4):
    est._update(step=0, occ_eval_fn=lambda x: field.query_density(x) * bench.RENDER_STEP, occ_thre=1e-2)
est.eval()

def frame_rays(k, W=800):
    g = torch.Generator().manual_seed(k)
    p = torch.randn(3, generator=g); p[2] = p[2].abs() * 0.7 + 0.2; p = 4.0 * p / p.norm()
    fwd = -p / p.norm(); up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up); right = right / right.norm(); tup = torch.linalg.cross(right, fwd)
    focal = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
    ys, xs = torch.meshgrid(torch.arange(W) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    d = fwd + ((xs - W / 2) / focal)[..., None] * right - ((ys - W / 2) / focal)[..., None] * tup
    d = d / d.norm(dim=-1, keepdim=True)
    return U.Rays(p.expand_as(d).contiguous().to(dev), d.contiguous().to(dev))

bk = torch.ones(3, device=dev)
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kw = dict(render_step_size=bench.RENDER_STEP, render_bkgd=bk)
methods = {
    "chunked 8192 (reference eval path)": lambda r: U.render_image_with_occgrid(field, est, r, test_chunk_size=8192, **kw),
    "chunked 65536": lambda r: U.render_image_with_occgrid(field, est, r, test_chunk_size=65536, **kw),
    "one chunk (640000 rays)": lambda r: U.render_image_with_occgrid(field, est, r, test_chunk_size=1 << 30, **kw),
    "test-mode marcher, reference API": lambda r: U.render_image_with_occgrid_test(1024, field, est, r, **kw),
    "test-mode marcher, fused rounds": lambda r: U.render_image_with_occgrid_test_fused(1024, field, est, r, **kw),
}
rays = [frame_rays(k) for k in range(n_frames)]
ref_img = None
print(f"| renderer | ms / 800x800 frame | M rays/s | M samples/frame | PSNR vs first row |")
print("|---|---|---|---|---|")
with torch.no_grad():
    for name, fn in methods.items():
        fn(rays[0]); torch.cuda.synchronize()
        t0 = time.perf_counter(); ns = 0
        for r in rays:
            img, opa, dep, n = fn(r); ns += n
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n_frames * 1e3
        img0 = fn(rays[0])[0]
        if ref_img is None:
            ref_img = img0
        mse = torch.mean((img0 - ref_img) ** 2).item()
        psnr = float("inf") if mse == 0 else -10 * math.log10(mse)
        print(f"| {name} | {ms:.1f} | {640000 / ms / 1e3:.2f} | {ns / n_frames / 1e6:.2f} | {psnr:.1f} |")
