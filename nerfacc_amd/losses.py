"""Mip-NeRF 360 distortion regulariser — nerfacc/losses.py:7-41."""
from torch import Tensor

from .scan import exclusive_sum
from .volrend import accumulate_along_rays


def distortion(weights: Tensor, t_starts: Tensor, t_ends: Tensor, ray_indices: Tensor, n_rays: int) -> Tensor:
    """Per-ray distortion loss (n_rays, 1) from flattened weights and intervals: the
    intra-interval term sum w^2 delta / 3 plus the pairwise term
    2 sum_i w_i (m_i sum_{j<i} w_j - sum_{j<i} w_j m_j), both per ray."""
    assert weights.shape == t_starts.shape == t_ends.shape == ray_indices.shape, (
        f"the shape of the inputs are not the same: weights {weights.shape}, t_starts {t_starts.shape}, "
        f"t_ends {t_ends.shape}, ray_indices {ray_indices.shape}")
    mids = 0.5 * (t_starts + t_ends)
    widths = t_ends - t_starts
    intra = (1 / 3) * (widths * weights.pow(2))
    w_before = exclusive_sum(weights, indices=ray_indices)
    wm_before = exclusive_sum(weights * mids, indices=ray_indices)
    pairwise = 2 * (weights * mids * w_before - weights * wm_before)
    return accumulate_along_rays(intra + pairwise, None, ray_indices, n_rays)
