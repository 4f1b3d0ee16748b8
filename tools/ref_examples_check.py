"""Drive the REFERENCE's examples/utils.py renderers (byte-identical copy under build/ref_suite/,
see tools/run_reference_suite.sh) on bench.py's scene through the `nerfacc` alias, and compare the
pixels with this repository's own renderers (examples/utils.py).

  render_image_with_occgrid        reference examples/utils.py:54-167  (eval: 8192-ray chunks; train: one
                                   chunk, stratified, backward)
  render_image_with_occgrid_test   :267-439 (iterative marcher: over-allocated traverse_grids,
                                   prefix_trans, accumulate_along_rays_)
  render_image_with_propnet        :170-264 (PropNetEstimator.sampling + batched rendering)

Prints a markdown table; exit code 1 if any row fails.
"""
import importlib.util
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITE = os.path.join(ROOT, "build", "ref_suite", "examples")
sys.path.insert(0, ROOT)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_utils():
    """the reference module does `from datasets.utils import Rays, namedtuple_map`: its own
    examples/datasets package must win over any installed `datasets` distribution."""
    for k in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.")]:
        del sys.modules[k]
    sys.path.insert(0, SUITE)
    try:
        return _load("reference_examples_utils", os.path.join(SUITE, "utils.py"))
    finally:
        sys.path.remove(SUITE)


def frame_rays(Rays, k, W, dev):
    g = torch.Generator().manual_seed(k)
    p = torch.randn(3, generator=g)
    p[2] = p[2].abs() * 0.7 + 0.2
    p = 4.0 * p / p.norm()
    fwd = -p / p.norm()
    up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    tup = torch.linalg.cross(right, fwd)
    focal = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
    ys, xs = torch.meshgrid(torch.arange(W) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    d = fwd + ((xs - W / 2) / focal)[..., None] * right - ((ys - W / 2) / focal)[..., None] * tup
    d = d / d.norm(dim=-1, keepdim=True)
    return Rays(p.expand_as(d).contiguous().to(dev), d.contiguous().to(dev))


class BatchedField(torch.nn.Module):
    """bench.DenseGridField for positions of any leading shape (the propnet renderer hands over
    [n_rays, n_samples, 3])."""

    def __init__(self, field):
        super().__init__()
        self.field = field

    def forward(self, x, dirs=None):
        rgb, sigma = self.field(x.reshape(-1, 3))
        return rgb.reshape(*x.shape[:-1], 3), sigma.reshape(*x.shape[:-1], 1)


class BatchedDensity(torch.nn.Module):
    def __init__(self, field, scale):
        super().__init__()
        self.field, self.scale = field, scale

    def forward(self, x):
        return self.field.query_density(x.reshape(-1, 3)).reshape(*x.shape[:-1], 1) * self.scale


def main():
    import bench
    import nerfacc

    R = load_reference_utils()
    U = _load("repo_examples_utils", os.path.join(ROOT, "examples", "utils.py"))
    assert nerfacc.OccGridEstimator is R.OccGridEstimator, "the reference module bound another estimator class"
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    field = bench.DenseGridField(bench.AABB, 128).to(dev)
    rows, ok_all = [], True

    def row(name, ok, detail):
        nonlocal ok_all
        ok_all &= bool(ok)
        rows.append(f"| {name} | {'PASS' if ok else 'FAIL'} | {detail} |")

    def maxdiff(a, b):
        return max(float((x - y).abs().max()) for x, y in zip(a[:3], b[:3]))

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    for levels, cone in ((1, 0.0), (2, 0.004)):
        est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=levels).to(dev)
        est.train()
        for _ in range(4):
            est._update(step=0, occ_eval_fn=lambda x: field.query_density(x) * bench.RENDER_STEP, occ_thre=1e-2)
        bk = torch.ones(3, device=dev)
        kw = dict(render_step_size=bench.RENDER_STEP, render_bkgd=bk, cone_angle=cone)
        tag = f"levels={levels}, cone_angle={cone}"

        # ---- eval, 8192-ray chunks, 800x800 frame -----------------------------------------
        field.eval()
        est.eval()
        rays_ref = frame_rays(R.Rays, 0, 800, dev)
        rays_own = U.Rays(rays_ref.origins, rays_ref.viewdirs)
        with torch.no_grad():
            ms_r, a = timed(lambda: R.render_image_with_occgrid(field, est, rays_ref, **kw))
            ms_o, b = timed(lambda: U.render_image_with_occgrid(field, est, rays_own, **kw))
        d = maxdiff(a, b)
        row(f"`render_image_with_occgrid` eval 800x800, 8192-ray chunks ({tag})", d <= 1e-5 and a[3] == b[3],
            f"max abs diff {d:.1e}; {a[3]} samples both; reference's loop {ms_r:.1f} ms / frame, this repo's {ms_o:.1f} ms")

        for at in (0.0, 1e-2):
            with torch.no_grad():
                c = R.render_image_with_occgrid(field, est, rays_ref, alpha_thre=at, **kw)
                ms_r, t = timed(lambda: R.render_image_with_occgrid_test(1024, field, est, rays_ref, alpha_thre=at, **kw))
                ms_o, u = timed(lambda: U.render_image_with_occgrid_test(1024, field, est, rays_own, alpha_thre=at, **kw))
                f = U.render_image_with_occgrid_test_fused(1024, field, est, rays_own, alpha_thre=at, **kw)
            d, df = maxdiff(t, u), maxdiff(t, f)
            mse = float(torch.mean((t[0] - c[0]) ** 2))
            psnr = float("inf") if mse == 0 else -10 * math.log10(mse)
            row(f"`render_image_with_occgrid_test` 800x800, alpha_thre={at} ({tag})",
                d <= 1e-5 and df <= 1e-4 and t[3] == u[3] and psnr > 35,
                f"vs this repo's marcher {d:.1e} ({t[3]} samples both), vs fused rounds {df:.1e}; PSNR vs the chunked image "
                f"{psnr:.1f} dB; reference's loop {ms_r:.1f} ms / frame, this repo's {ms_o:.1f} ms")

        # ---- training step: one chunk, stratified, backward -------------------------------
        field.train()
        est.train()
        pool_o, pool_d = bench.make_ray_pool(8192, 7, dev)
        grads = []
        for mod, rays in ((R, R.Rays(pool_o, pool_d)), (U, U.Rays(pool_o, pool_d))):
            torch.manual_seed(123)
            field.zero_grad(set_to_none=True)
            rgb, opa, dep, n = mod.render_image_with_occgrid(field, est, rays, alpha_thre=1e-2, **kw)
            (rgb.square().mean() + dep.mean()).backward()
            grads.append((rgb.detach(), opa.detach(), dep.detach(), n, field.grid.grad.clone()))
        d = maxdiff(grads[0], grads[1])
        dg = float((grads[0][4] - grads[1][4]).abs().max())
        gn = float(grads[0][4].abs().max())
        row(f"`render_image_with_occgrid` training step, 8192 rays, stratified + backward ({tag})",
            d <= 1e-5 and dg <= 1e-3 * gn and grads[0][3] == grads[1][3] and math.isfinite(gn) and gn > 0,
            f"outputs {d:.1e}, field gradient {dg:.1e} (max |g| {gn:.2e}), {grads[0][3]} samples both")

    # ---- PropNet renderer (reference examples/utils.py:170-264) ---------------------------
    field.train()
    prop_nets = [BatchedDensity(field, 0.5).to(dev), BatchedDensity(field, 0.8).to(dev)]
    pest = nerfacc.PropNetEstimator().to(dev)
    pool_o, pool_d = bench.make_ray_pool(4096, 11, dev)
    rays = R.Rays(pool_o, pool_d)
    torch.manual_seed(5)
    field.zero_grad(set_to_none=True)
    rgb, opa, dep, extras = R.render_image_with_propnet(
        BatchedField(field), prop_nets, pest, rays, num_samples=48, num_samples_per_prop=[256, 96],
        near_plane=2.0, far_plane=6.0, sampling_type="uniform", opaque_bkgd=False,
        render_bkgd=torch.ones(3, device=dev), proposal_requires_grad=True)
    trans = extras["trans"] if isinstance(extras, dict) and "trans" in extras else None
    loss = rgb.square().mean()
    if trans is not None:
        loss = loss + pest.compute_loss(trans.detach(), loss_scaler=1.0)
    loss.backward()
    g = field.grid.grad
    ok = bool(torch.isfinite(rgb).all() and torch.isfinite(g).all() and g.abs().max() > 0 and rgb.shape == (4096, 3)
              and float(opa.min()) >= 0 and float(opa.max()) <= 1 + 1e-5)
    row("`render_image_with_propnet` training step 4096 x (256, 96) -> 48, `compute_loss` + backward", ok,
        f"rgb {tuple(rgb.shape)}, opacity in [{float(opa.min()):.3f}, {float(opa.max()):.3f}], max |grad| {float(g.abs().max()):.2e}")
    field.eval()
    with torch.no_grad():
        a = R.render_image_with_propnet(
            BatchedField(field), prop_nets, pest, frame_rays(R.Rays, 1, 200, dev), num_samples=48,
            num_samples_per_prop=[256, 96], near_plane=2.0, far_plane=6.0, sampling_type="lindisp", opaque_bkgd=True)
    row("`render_image_with_propnet` eval 200x200, 8192-ray chunks, lindisp, opaque background",
        bool(torch.isfinite(a[0]).all()) and a[0].shape == (200, 200, 3) and float(a[1].min()) > 0.999,
        f"opacity min {float(a[1].min()):.4f} (opaque background), colours finite")

    print("| reference renderer, unmodified | result | detail |")
    print("|---|---|---|")
    print("\n".join(rows))
    print()
    print(f"device: {torch.cuda.get_device_name(0)}; torch {torch.__version__}")
    return 0 if ok_all else 1



if __name__ == "__main__":
    sys.exit(main())
