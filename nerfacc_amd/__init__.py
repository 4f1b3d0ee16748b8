"""nerfacc_amd — MI355X-native implementation of nerfacc's OccGrid sampling + volumetric
rendering path, behind nerfacc's own Python API (nerfacc/__init__.py:5-24).

    import nerfacc_amd as nerfacc      # or `import nerfacc` via the alias package at the repo root

Native code: nerfacc_amd/libnerfacc_hip.so (hand-written HIP for gfx950, C ABI in
include/nerfacc_hip.h), built by `python -m nerfacc_amd.build`.
"""
from .data_specs import RayIntervals, RaySamples
from .estimators.occ_grid import OccGridEstimator
from .estimators.prop_net import PropNetEstimator
from .grid import ray_aabb_intersect, sample_positions, traverse_grids
from .losses import distortion
from .options import get_option, list_options, options, release_workspace, reset_options, set_option
from .pack import pack_info
from .pdf import importance_sampling, searchsorted
from .scan import exclusive_prod, exclusive_sum, inclusive_prod, inclusive_sum
from .version import __version__
from .volrend import (
    accumulate_along_rays,
    render_transmittance_from_alpha,
    render_transmittance_from_density,
    render_visibility_from_alpha,
    render_visibility_from_density,
    render_weight_from_alpha,
    render_weight_from_density,
    rendering,
)

__all__ = [
    "__version__",
    "inclusive_prod", "exclusive_prod", "inclusive_sum", "exclusive_sum",
    "pack_info",
    "render_visibility_from_alpha", "render_visibility_from_density",
    "render_weight_from_alpha", "render_weight_from_density",
    "render_transmittance_from_alpha", "render_transmittance_from_density",
    "accumulate_along_rays", "rendering",
    "importance_sampling", "searchsorted",
    "RayIntervals", "RaySamples",
    "ray_aabb_intersect", "traverse_grids",
    "OccGridEstimator", "PropNetEstimator",
    "distortion",
]
# additions of this implementation (not in the reference's list of 24 names)
__all__ += ["sample_positions", "set_option", "get_option", "reset_options", "list_options", "options", "release_workspace"]
