"""Image renderers on top of nerfacc_amd, with the call signatures of the reference's
examples/utils.py (render_image_with_occgrid :54-167, render_image_with_occgrid_test :267-439),
so that a training script written against the reference keeps working.

Three renderers:
  render_image_with_occgrid         chunked sampling + rendering (training and chunked eval)
  render_image_with_occgrid_test    the test-time iterative marcher, composed from the
                                    reference-compatible API (over-allocated traverse_grids with
                                    rays_mask / termination planes, prefix_trans,
                                    accumulate_along_rays_)
  render_image_with_occgrid_test_fused
                                    the same marcher with each round's samples produced exactly
                                    sized and compacted by one fused call (no over-allocation, no
                                    boolean-mask gathers): same pixels, fewer host syncs.

A radiance field is any module with  forward(positions[N,3], dirs[N,3]) -> (rgb[N,3], sigma[N,1])
and  query_density(positions) -> sigma[N,1].
"""
import collections
from typing import Optional

import torch

import nerfacc_amd as nerfacc
from nerfacc_amd import cuda as _C
from nerfacc_amd.grid import ray_aabb_intersect, traverse_grids
from nerfacc_amd.volrend import accumulate_along_rays_, render_weight_from_density, rendering

Rays = collections.namedtuple("Rays", ("origins", "viewdirs"))


def _flatten(rays: Rays):
    shape = rays.origins.shape
    if len(shape) == 3:
        h, w, _ = shape
        rays = Rays(rays.origins.reshape(h * w, 3), rays.viewdirs.reshape(h * w, 3))
    return rays, shape


def render_image_with_occgrid(
    radiance_field, estimator, rays: Rays, near_plane: float = 0.0, far_plane: float = 1e10,
    render_step_size: float = 1e-3, render_bkgd: Optional[torch.Tensor] = None, cone_angle: float = 0.0,
    alpha_thre: float = 0.0, test_chunk_size: int = 8192,
):
    """examples/utils.py:54-167 — returns (rgb, opacity, depth, n_rendering_samples)."""
    rays, shape = _flatten(rays)
    num_rays = rays.origins.shape[0]
    results = []
    chunk = torch.iinfo(torch.int32).max if radiance_field.training else test_chunk_size
    for i in range(0, num_rays, chunk):
        o, d = rays.origins[i:i + chunk], rays.viewdirs[i:i + chunk]

        def sigma_fn(t_starts, t_ends, ray_indices):
            if t_starts.shape[0] == 0:
                return torch.empty((0,), device=t_starts.device)
            pos = o[ray_indices] + d[ray_indices] * ((t_starts + t_ends)[:, None] / 2.0)
            return radiance_field.query_density(pos).squeeze(-1)

        def rgb_sigma_fn(t_starts, t_ends, ray_indices):
            if t_starts.shape[0] == 0:
                return torch.empty((0, 3), device=t_starts.device), torch.empty((0,), device=t_starts.device)
            dirs = d[ray_indices]
            pos = o[ray_indices] + dirs * ((t_starts + t_ends)[:, None] / 2.0)
            rgbs, sigmas = radiance_field(pos, dirs)
            return rgbs, sigmas.squeeze(-1)

        ray_indices, t_starts, t_ends = estimator.sampling(
            o, d, sigma_fn=sigma_fn, near_plane=near_plane, far_plane=far_plane, render_step_size=render_step_size,
            stratified=radiance_field.training, cone_angle=cone_angle, alpha_thre=alpha_thre)
        rgb, opacity, depth, _ = rendering(t_starts, t_ends, ray_indices, n_rays=o.shape[0],
                                           rgb_sigma_fn=rgb_sigma_fn, render_bkgd=render_bkgd)
        results.append((rgb, opacity, depth, t_starts.shape[0]))
    colors = torch.cat([r[0] for r in results]).view((*shape[:-1], -1))
    opacities = torch.cat([r[1] for r in results]).view((*shape[:-1], -1))
    depths = torch.cat([r[2] for r in results]).view((*shape[:-1], -1))
    return colors, opacities, depths, sum(r[3] for r in results)


def _finish(rgb, opacity, depth, render_bkgd, shape, total):
    if render_bkgd is not None:
        rgb = rgb + render_bkgd * (1.0 - opacity)
    depth = depth / opacity.clamp_min(torch.finfo(torch.float32).eps)
    return rgb.view((*shape[:-1], -1)), opacity.view((*shape[:-1], -1)), depth.view((*shape[:-1], -1)), total


@torch.no_grad()
def render_image_with_occgrid_test(
    max_samples: int, radiance_field, estimator, rays: Rays, near_plane: float = 0.0, far_plane: float = 1e10,
    render_step_size: float = 1e-3, render_bkgd: Optional[torch.Tensor] = None, cone_angle: float = 0.0,
    alpha_thre: float = 0.0, early_stop_eps: float = 1e-4,
):
    """examples/utils.py:267-439: march all rays a few samples at a time, composite, drop the rays
    that became opaque or left the grid, repeat."""
    rays, shape = _flatten(rays)
    rays_o, rays_d = rays.origins.contiguous(), rays.viewdirs.contiguous()
    num_rays, device = rays_o.shape[0], rays_o.device
    opacity = torch.zeros(num_rays, 1, device=device)
    depth = torch.zeros(num_rays, 1, device=device)
    rgb = torch.zeros(num_rays, 3, device=device)
    ray_mask = torch.ones(num_rays, device=device).bool()
    min_samples = 1 if cone_angle == 0 else 4
    iter_samples = total_samples = 0
    near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
    far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
    t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, estimator.aabbs)
    n_grids = estimator.binaries.size(0)
    if n_grids > 1:
        t_sorted, t_indices = torch.sort(torch.cat([t_mins, t_maxs], -1), -1)
    else:
        t_sorted = torch.cat([t_mins, t_maxs], -1)
        t_indices = torch.arange(0, n_grids * 2, device=device, dtype=torch.int64).expand(num_rays, n_grids * 2)
    opc_thre = 1 - early_stop_eps
    while iter_samples < max_samples:
        n_alive = ray_mask.sum().item()
        if n_alive == 0:
            break
        n_samples = max(min(num_rays // n_alive, 64), min_samples)
        iter_samples += n_samples
        intervals, samples, termination_planes = traverse_grids(
            rays_o, rays_d, estimator.binaries, estimator.aabbs, near_planes, far_planes, render_step_size, cone_angle,
            n_samples, True, ray_mask, t_sorted, t_indices.contiguous(), hits)
        t_starts = intervals.vals[intervals.is_left]
        t_ends = intervals.vals[intervals.is_right]
        ray_indices = samples.ray_indices[samples.is_valid]
        packed_info = samples.packed_info
        dirs = rays_d[ray_indices]
        rgbs, sigmas = radiance_field(rays_o[ray_indices] + dirs * ((t_starts + t_ends)[:, None] / 2.0), dirs)
        weights, _, alphas = render_weight_from_density(
            t_starts, t_ends, sigmas.squeeze(-1), ray_indices=ray_indices, n_rays=num_rays,
            prefix_trans=1 - opacity[ray_indices].squeeze(-1))
        if alpha_thre > 0:
            vis = alphas >= alpha_thre
            ray_indices, rgbs, weights, t_starts, t_ends = ray_indices[vis], rgbs[vis], weights[vis], t_starts[vis], t_ends[vis]
        accumulate_along_rays_(weights, values=rgbs, ray_indices=ray_indices, outputs=rgb)
        accumulate_along_rays_(weights, values=None, ray_indices=ray_indices, outputs=opacity)
        accumulate_along_rays_(weights, values=(t_starts + t_ends)[..., None] / 2.0, ray_indices=ray_indices, outputs=depth)
        # (skipped rays keep their plane: the kernel only writes the planes of the rays it marched)
        near_planes = torch.where(ray_mask, termination_planes, near_planes)
        ray_mask = torch.logical_and(opacity.view(-1) <= opc_thre, packed_info[:, 1] == n_samples)
        total_samples += ray_indices.shape[0]
    return _finish(rgb, opacity, depth, render_bkgd, shape, total_samples)


@torch.no_grad()
def render_image_with_occgrid_test_fused(
    max_samples: int, radiance_field, estimator, rays: Rays, near_plane: float = 0.0, far_plane: float = 1e10,
    render_step_size: float = 1e-3, render_bkgd: Optional[torch.Tensor] = None, cone_angle: float = 0.0,
    alpha_thre: float = 0.0, early_stop_eps: float = 1e-4,
):
    """Same marcher, each round through nerfacc_amd.cuda.sample_occgrid(rays_mask=...,
    traverse_steps_limit=...): the round's samples arrive exactly sized and compacted (count pass,
    offsets, emit pass) — no over-allocated buffers, no is_left / is_right / is_valid gathers."""
    rays, shape = _flatten(rays)
    rays_o, rays_d = rays.origins.contiguous(), rays.viewdirs.contiguous()
    num_rays, device = rays_o.shape[0], rays_o.device
    opacity = torch.zeros(num_rays, 1, device=device)
    depth = torch.zeros(num_rays, 1, device=device)
    rgb = torch.zeros(num_rays, 3, device=device)
    ray_mask = torch.ones(num_rays, device=device).bool()
    min_samples = 1 if cone_angle == 0 else 4
    iter_samples = total_samples = 0
    near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
    far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
    binaries, aabbs = estimator.binaries.contiguous(), estimator.aabbs.contiguous()
    opc_thre = 1 - early_stop_eps
    n_alive = num_rays
    while iter_samples < max_samples and n_alive > 0:
        n_samples = max(min(num_rays // n_alive, 64), min_samples)
        iter_samples += n_samples
        ray_indices, t_starts, t_ends, packed_info, near_planes = _C.sample_occgrid(
            rays_o, rays_d, binaries, aabbs, near_planes, far_planes, render_step_size, cone_angle,
            rays_mask=ray_mask, traverse_steps_limit=n_samples, with_terminate_planes=True)
        if ray_indices.shape[0] > 0:
            dirs = rays_d[ray_indices]
            rgbs, sigmas = radiance_field(rays_o[ray_indices] + dirs * ((t_starts + t_ends)[:, None] / 2.0), dirs)
            weights, _, alphas = render_weight_from_density(
                t_starts, t_ends, sigmas.squeeze(-1), ray_indices=ray_indices, n_rays=num_rays,
                prefix_trans=1 - opacity[ray_indices].squeeze(-1))
            if alpha_thre > 0:
                vis = alphas >= alpha_thre
                ray_indices, rgbs, weights, t_starts, t_ends = ray_indices[vis], rgbs[vis], weights[vis], t_starts[vis], t_ends[vis]
            accumulate_along_rays_(weights, values=rgbs, ray_indices=ray_indices, outputs=rgb)
            accumulate_along_rays_(weights, values=None, ray_indices=ray_indices, outputs=opacity)
            accumulate_along_rays_(weights, values=(t_starts + t_ends)[..., None] / 2.0, ray_indices=ray_indices, outputs=depth)
            total_samples += ray_indices.shape[0]
        ray_mask = torch.logical_and(opacity.view(-1) <= opc_thre, packed_info[:, 1] == n_samples)
        n_alive = int(ray_mask.sum().item())
    return _finish(rgb, opacity, depth, render_bkgd, shape, total_samples)
