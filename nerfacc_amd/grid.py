"""Ray/AABB intersection and multi-level occupancy-grid traversal — nerfacc/grid.py."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import cuda as _C
from .data_specs import RayIntervals, RaySamples


@torch.no_grad()
def ray_aabb_intersect(rays_o: Tensor, rays_d: Tensor, aabbs: Tensor, near_plane: float = -float("inf"),
                       far_plane: float = float("inf"), miss_value: float = float("inf")
                       ) -> Tuple[Tensor, Tensor, Tensor]:
    """Slab test of every ray against every box (grid.py:13-51).

    rays_o, rays_d: (n_rays, 3); aabbs: (m, 6) as {xmin, ymin, zmin, xmax, ymax, zmax}.
    Returns t_mins, t_maxs (n_rays, m), clipped to [near_plane, far_plane] and set to
    `miss_value` where the ray misses, and the boolean hits (n_rays, m).
    """
    assert rays_o.ndim == 2 and rays_o.shape[-1] == 3
    assert rays_d.ndim == 2 and rays_d.shape[-1] == 3
    assert aabbs.ndim == 2 and aabbs.shape[-1] == 6
    t_mins, t_maxs, hits = _C.ray_aabb_intersect(rays_o.contiguous(), rays_d.contiguous(), aabbs.contiguous(),
                                                 near_plane, far_plane, miss_value)
    return t_mins, t_maxs, hits


def _ray_aabb_intersect(rays_o: Tensor, rays_d: Tensor, aabbs: Tensor, near_plane: float = -float("inf"),
                        far_plane: float = float("inf"), miss_value: float = float("inf")
                        ) -> Tuple[Tensor, Tensor, Tensor]:
    """Pure-torch twin of :func:`ray_aabb_intersect` (grid.py:54-90); runs on any device."""
    lo, hi = aabbs[None, :, :3], aabbs[None, :, 3:]
    o, d = rays_o[:, None, :], rays_d[:, None, :]
    ta, tb = (lo - o) / d, (hi - o) / d
    t_mins = torch.minimum(ta, tb).amax(dim=-1)
    t_maxs = torch.maximum(ta, tb).amin(dim=-1)
    hits = (t_maxs > t_mins) & (t_maxs > 0)
    t_mins = torch.where(hits, t_mins.clamp(near_plane, far_plane), miss_value)
    t_maxs = torch.where(hits, t_maxs.clamp(near_plane, far_plane), miss_value)
    return t_mins, t_maxs, hits


@torch.no_grad()
def traverse_grids(
    rays_o: Tensor, rays_d: Tensor, binaries: Tensor, aabbs: Tensor,
    near_planes: Optional[Tensor] = None, far_planes: Optional[Tensor] = None,
    step_size: Optional[float] = 1e-3, cone_angle: Optional[float] = 0.0,
    traverse_steps_limit: Optional[int] = None, over_allocate: Optional[bool] = False,
    rays_mask: Optional[Tensor] = None,
    t_sorted: Optional[Tensor] = None, t_indices: Optional[Tensor] = None, hits: Optional[Tensor] = None,
) -> Tuple[RayIntervals, RaySamples, Tensor]:
    """March rays through one or more nested occupancy grids (grid.py:93-192).

    binaries: (m, rx, ry, rz) bool, aabbs: (m, 6); earlier grids win where grids overlap.
    near/far_planes: per-ray limits (default 0 / inf).  step_size > 0 marches on a lattice
    (growing with `cone_angle`), step_size <= 0 emits one interval per occupied voxel.
    `traverse_steps_limit` caps samples per ray; with `over_allocate` the outputs use fixed
    slots of that size (test-time iterative marching) and `rays_mask` selects live rays.
    Pre-computed `t_sorted`/`t_indices`/`hits` (sorted ray-grid entry/exit events) may be
    passed; otherwise the kernel derives them per ray itself.

    Returns (RayIntervals, RaySamples, terminate_planes[n_rays]).  Not differentiable.
    """
    if near_planes is None:
        near_planes = torch.zeros_like(rays_o[:, 0])
    if far_planes is None:
        far_planes = torch.full_like(rays_o[:, 0], float("inf"))
    if rays_mask is None:
        rays_mask = torch.ones_like(rays_o[:, 0], dtype=torch.bool)
    if traverse_steps_limit is None:
        traverse_steps_limit = -1
    if over_allocate:
        assert traverse_steps_limit > 0, "traverse_steps_limit must be set if over_allocate is True."
    have_events = t_sorted is not None and t_indices is not None and hits is not None
    intervals, samples, termination_planes = _C.traverse_grids(
        rays_o.contiguous(), rays_d.contiguous(), rays_mask.contiguous(),
        binaries.contiguous(), aabbs.contiguous(),
        t_sorted.contiguous() if have_events else None,
        t_indices.contiguous() if have_events else None,
        hits.contiguous() if have_events else None,
        near_planes.contiguous(), far_planes.contiguous(),
        step_size, cone_angle, True, True, True, traverse_steps_limit, over_allocate,
    )
    return RayIntervals._from_cpp(intervals), RaySamples._from_cpp(samples), termination_planes


def _enlarge_aabb(aabb, factor: float) -> Tensor:
    centre = (aabb[:3] + aabb[3:]) / 2
    half = (aabb[3:] - aabb[:3]) / 2
    return torch.cat([centre - half * factor, centre + half * factor])


def _query(x: Tensor, data: Tensor, base_aabb: Tensor) -> Tuple[Tensor, Tensor]:
    """Look up nested-grid values at points (grid.py:201-237): level i covers the base box
    scaled by 2**i; a point uses the finest level that contains it.  Returns (values masked
    by validity, validity)."""
    lo, hi = base_aabb[:3], base_aabb[3:]
    unit = (x - lo) / (hi - lo)
    # the exponent of the largest |offset from centre| selects the level; clamping at 0.1
    # keeps frexp away from 0
    reach = (unit - 0.5).abs().amax(dim=-1).clamp(min=0.1)
    mip = (torch.frexp(reach)[1].long() + 1).clamp(min=0)
    inside = mip < data.shape[0]
    unit_lvl = (unit - 0.5) / (2**mip)[:, None] + 0.5
    res = torch.tensor(data.shape[1:], device=x.device)
    idx = torch.minimum((unit_lvl * res).long(), res - 1)
    mip = mip.clamp(max=data.shape[0] - 1)
    return data[mip, idx[:, 0], idx[:, 1], idx[:, 2]] * inside, inside


def sample_positions(rays_o: Tensor, rays_d: Tensor, ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor,
                     return_dirs: bool = False):
    """World-space midpoints of flattened samples (an addition of this implementation):

        positions = rays_o[ray_indices] + rays_d[ray_indices] * ((t_starts + t_ends)[:, None] / 2)

    the first line of every `sigma_fn` / `rgb_sigma_fn` in the reference's examples
    (examples/utils.py:96-101), as one launch instead of six, bit-identical to the expression.
    `return_dirs` also returns `rays_d[ray_indices]`.  If an input requires grad (pose or ray
    optimisation) the torch expression itself is evaluated so that autograd sees it.
    """
    needs_grad = torch.is_grad_enabled() and any(x.requires_grad for x in (rays_o, rays_d, t_starts, t_ends))
    if needs_grad:
        dirs = rays_d[ray_indices]
        pos = rays_o[ray_indices] + dirs * ((t_starts + t_ends)[:, None] / 2.0)
        return (pos, dirs) if return_dirs else pos
    return _C.sample_positions(rays_o.contiguous(), rays_d.contiguous(), ray_indices.contiguous(),
                               t_starts.contiguous(), t_ends.contiguous(), return_dirs)
