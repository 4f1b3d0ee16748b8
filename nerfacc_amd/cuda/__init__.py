"""`nerfacc_amd.cuda` — same role and names as the reference's `nerfacc.cuda`
(nerfacc/cuda/__init__.py:8-53): thin, lazily bound accessors onto the native backend.

Resolution is deferred to the first call, so importing the package (e.g. to use the batched,
pure-torch code paths or to build the library) does not require libnerfacc_hip.so yet; the
first native call does, and raises if it is absent.  "cuda" is kept as the module name because
it is the torch device type on ROCm and what callers of the reference import.
"""
from typing import Callable


def _lazy(name: str) -> Callable:
    def call(*args, **kwargs):
        from ._backend import _C  # noqa: WPS433 (deferred on purpose)

        return getattr(_C, name)(*args, **kwargs)

    call.__name__ = name
    return call


# the 21 names of the reference boundary (nerfacc.cpp:126-163)
_REFERENCE_NAMES = (
    "RaySegmentsSpec",
    "ray_aabb_intersect", "traverse_grids",
    "inclusive_sum", "exclusive_sum",
    "inclusive_prod_forward", "inclusive_prod_backward",
    "exclusive_prod_forward", "exclusive_prod_backward",
    "is_cub_available",
    "inclusive_sum_cub", "exclusive_sum_cub",
    "inclusive_prod_cub_forward", "inclusive_prod_cub_backward",
    "exclusive_prod_cub_forward", "exclusive_prod_cub_backward",
    "importance_sampling", "searchsorted",
    "opencv_lens_undistortion", "opencv_lens_undistortion_fisheye",
)
# fused entry points added by this implementation (see _backend._C)
_FUSED_NAMES = (
    "sample_occgrid", "pack_info", "unpack_info",
    "render_weight_from_density_fwd", "render_weight_from_density_bwd",
    "visibility_compact", "accumulate_along_rays", "accumulate_along_rays_bwd",
    "rendering_fwd", "rendering_bwd",
    "grid_cell_points", "grid_ema_update", "grid_threshold", "grid_mark_invisible", "grid_occupied_counts", "grid_occupied_cells", "sample_positions",
    "transform_stot", "edge_cdfs_fwd", "edge_cdfs_bwd", "pdf_loss_fwd", "pdf_loss_bwd",
)

for _n in _REFERENCE_NAMES + _FUSED_NAMES:
    globals()[_n] = _lazy(_n)
del _n

__all__ = list(_REFERENCE_NAMES + _FUSED_NAMES)
