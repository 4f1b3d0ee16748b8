"""Test-time iterative marching (examples/utils.py:267-439) built only from the
public API: over-allocated traverse_grids with rays_mask + termination planes,
render_weight_from_density(prefix_trans=...), in-place accumulate_along_rays_.  It must agree
with the training-path rendering of the same rays."""
import numpy as np
import pytest
import torch

from gpu_utils import DEV, lego_like, t

pytestmark = pytest.mark.gpu


def _field(pos):
    sigma = 40.0 * torch.exp(-6.0 * (pos.norm(dim=-1) - 0.7) ** 2)
    rgb = torch.sigmoid(3.0 * pos)
    return rgb, sigma


@torch.no_grad()
def _render_test_mode(est, rays_o, rays_d, render_step_size, bkgd, max_samples=1024, early_stop_eps=1e-4):
    from nerfacc_amd.grid import ray_aabb_intersect, traverse_grids
    from nerfacc_amd.volrend import accumulate_along_rays_, render_weight_from_density

    R = rays_o.shape[0]
    opacity = torch.zeros(R, 1, device=DEV)
    depth = torch.zeros(R, 1, device=DEV)
    rgb = torch.zeros(R, 3, device=DEV)
    ray_mask = torch.ones(R, device=DEV).bool()
    near_planes = torch.zeros(R, device=DEV)
    far_planes = torch.full((R,), 1e10, device=DEV)
    t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, est.aabbs)
    t_sorted = torch.cat([t_mins, t_maxs], -1)                      # one level: already sorted
    t_indices = torch.arange(0, 2, device=DEV, dtype=torch.int64).expand(R, 2).contiguous()
    iter_samples = total = 0
    while iter_samples < max_samples:
        n_alive = int(ray_mask.sum().item())
        if n_alive == 0:
            break
        n_samples = max(min(R // n_alive, 64), 1)
        iter_samples += n_samples
        intervals, samples, term = traverse_grids(rays_o, rays_d, est.binaries, est.aabbs, near_planes, far_planes,
                                                  render_step_size, 0.0, n_samples, True, ray_mask, t_sorted, t_indices, hits)
        t_starts = intervals.vals[intervals.is_left]
        t_ends = intervals.vals[intervals.is_right]
        ray_indices = samples.ray_indices[samples.is_valid]
        pos = rays_o[ray_indices] + rays_d[ray_indices] * ((t_starts + t_ends)[:, None] / 2.0)
        rgbs, sigmas = _field(pos)
        weights, _, _ = render_weight_from_density(t_starts, t_ends, sigmas, ray_indices=ray_indices, n_rays=R,
                                                   prefix_trans=1 - opacity[ray_indices].squeeze(-1))
        accumulate_along_rays_(weights, values=rgbs, ray_indices=ray_indices, outputs=rgb)
        accumulate_along_rays_(weights, values=None, ray_indices=ray_indices, outputs=opacity)
        accumulate_along_rays_(weights, values=(t_starts + t_ends)[..., None] / 2.0, ray_indices=ray_indices, outputs=depth)
        near_planes = torch.where(ray_mask, term, near_planes)
        ray_mask = torch.logical_and(opacity.view(-1) <= 1 - early_stop_eps, samples.packed_info[:, 1] == n_samples)
        total += ray_indices.shape[0]
    rgb = rgb + bkgd * (1.0 - opacity)
    depth = depth / opacity.clamp_min(torch.finfo(torch.float32).eps)
    return rgb, opacity, depth, total


def test_iterative_test_mode_matches_training_path():
    import nerfacc_amd as nerfacc

    o, d, aabb, occ = lego_like(9, 3000, res=64)
    est = nerfacc.OccGridEstimator(roi_aabb=t(aabb[0]), resolution=64, levels=1).to(DEV)
    est.binaries = t(occ)
    O, D = t(o), t(d)
    bk = torch.tensor([1.0, 1.0, 1.0], device=DEV)
    step = 1e-2
    rgb_t, opa_t, dep_t, n_test = _render_test_mode(est, O, D, step, bk)

    def sigma_fn(ts, te, ri):
        return _field(O[ri] + D[ri] * ((ts + te)[:, None] / 2.0))[1]

    def rgb_sigma_fn(ts, te, ri):
        return _field(O[ri] + D[ri] * ((ts + te)[:, None] / 2.0))

    ri, ts, te = est.sampling(O, D, sigma_fn=sigma_fn, render_step_size=step, early_stop_eps=1e-4)
    rgb, opa, dep, _ = nerfacc.rendering(ts, te, ri, n_rays=3000, rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bk)
    assert n_test > 0 and ri.shape[0] > 0
    # same lattice, same samples up to where each path stops early (T < 1e-4): colours agree
    assert torch.allclose(rgb_t, rgb, atol=2e-3), (rgb_t - rgb).abs().max()
    assert torch.allclose(opa_t, opa, atol=2e-3)
    hit = opa[:, 0] > 0.5
    assert torch.allclose(dep_t[hit], dep[hit], atol=2e-2)


def test_fused_rounds_match_reference_api_marcher():
    """examples/utils.py: the fused per-round call (exactly sized, compacted samples) must march
    the same samples as the over-allocated traverse_grids + mask gathers of the reference loop."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import nerfacc_amd as nerfacc
    import utils as U
    from nerfacc_amd import cuda as C
    from nerfacc_amd.grid import traverse_grids

    o, d, aabb, occ = lego_like(11, 4000, res=64)
    est = nerfacc.OccGridEstimator(roi_aabb=t(aabb[0]), resolution=64, levels=1).to(DEV)
    est.binaries = t(occ)
    O, D = t(o), t(d)

    class Field(torch.nn.Module):
        def forward(self, pos, dirs=None):
            rgb, sigma = _field(pos)
            return rgb, sigma[:, None]

    # one round, sample by sample: mask + per-ray limit
    mask = torch.rand(4000, device=DEV) < 0.6
    near = torch.rand(4000, device=DEV) * 0.3
    far = torch.full((4000,), 1e10, device=DEV)
    iv, sm, term = traverse_grids(O, D, est.binaries, est.aabbs, near, far, 1e-2, 0.0, 7, True, mask)
    ri, ts, te, pk, term_f = C.sample_occgrid(O, D, est.binaries, est.aabbs, near, far, 1e-2, 0.0, rays_mask=mask,
                                              traverse_steps_limit=7, with_terminate_planes=True)
    assert torch.equal(ri, sm.ray_indices[sm.is_valid])
    assert torch.equal(ts, iv.vals[iv.is_left]) and torch.equal(te, iv.vals[iv.is_right])
    assert torch.equal(pk[:, 1], sm.packed_info[:, 1]) and int(pk[:, 1].max()) == 7
    assert torch.equal(term_f[mask], term[mask]) and torch.equal(term_f[~mask], near[~mask])
    assert (pk[~mask, 1] == 0).all()

    bk = torch.ones(3, device=DEV)
    a = U.render_image_with_occgrid_test(1024, Field(), est, U.Rays(O, D), render_step_size=1e-2, render_bkgd=bk)
    b = U.render_image_with_occgrid_test_fused(1024, Field(), est, U.Rays(O, D), render_step_size=1e-2, render_bkgd=bk)
    assert a[3] == b[3] and a[3] > 0                                   # same number of marched samples
    for x, y in zip(a[:3], b[:3]):
        assert torch.allclose(x, y, atol=1e-5)                         # float atomics order only
