"""Per-ray inverse-transform sampling and searchsorted — nerfacc/pdf.py."""
from typing import Tuple, Union

import torch
from torch import Tensor

from . import cuda as _C
from .data_specs import RayIntervals, RaySamples


def searchsorted(sorted_sequence: Union[RayIntervals, RaySamples], values: Union[RayIntervals, RaySamples]
                 ) -> Tuple[Tensor, Tensor]:
    """For each value find (ids_left, ids_right) in the same ray's sorted sequence such that
    seq[ids_left] <= value < seq[ids_right]; values outside the sequence behave as if clipped to
    it (pdf.py:13-62).  Both arguments may be batched or flattened ray containers.

        >>> seq = RayIntervals(vals=torch.tensor([0., 1., 0., 1., 2.], device="cuda"),
        ...                    packed_info=torch.tensor([[0, 2], [2, 3]], device="cuda"))
        >>> q = RayIntervals(vals=torch.tensor([0.5, 1.5, 2.5], device="cuda"),
        ...                  packed_info=torch.tensor([[0, 1], [1, 2]], device="cuda"))
        >>> searchsorted(seq, q)
        (tensor([0, 3, 3]), tensor([1, 4, 4]))
    """
    ids_left, ids_right = _C.searchsorted(values._to_cpp(), sorted_sequence._to_cpp())
    return ids_left, ids_right


def importance_sampling(intervals: RayIntervals, cdfs: Tensor, n_intervals_per_ray: Union[Tensor, int],
                        stratified: bool = False) -> Tuple[RayIntervals, RaySamples]:
    """Resample each ray into `n_intervals_per_ray` intervals by inverting the cdf given at the
    edges of `intervals` (pdf.py:65-131).  With an int count the results are batched:
    intervals.vals (n_rays, n + 1) and samples.vals (n_rays, n).  `stratified` shifts the
    regular sampling positions by one uniform offset per ray.

        >>> iv = RayIntervals(vals=torch.tensor([0., 1., 0., 1., 2.], device="cuda"),
        ...                   packed_info=torch.tensor([[0, 2], [2, 3]], device="cuda"))
        >>> new_iv, s = importance_sampling(iv, torch.tensor([0., .5, 0., .5, 1.], device="cuda"), 2)
        >>> new_iv.vals, s.vals
        ([[0.0, 0.5, 1.0], [0.0, 1.0, 2.0]], [[0.25, 0.75], [0.5, 1.5]])
    """
    if isinstance(n_intervals_per_ray, Tensor):
        n_intervals_per_ray = n_intervals_per_ray.contiguous()
    new_intervals, samples = _C.importance_sampling(intervals._to_cpp(), cdfs.contiguous(), n_intervals_per_ray,
                                                    stratified)
    return RayIntervals._from_cpp(new_intervals), RaySamples._from_cpp(samples)


def _sample_from_weighted(bins: Tensor, weights: Tensor, num_samples: int, stratified: bool = False,
                          vmin: float = -torch.inf, vmax: float = torch.inf) -> Tuple[Tensor, Tensor]:
    """Pure-torch twin of :func:`importance_sampling` for batched histograms (pdf.py:134-219):
    bins (..., B + 1), weights (..., B) -> edges (..., S + 1), centres (..., S)."""
    n_bins = weights.shape[-1]
    assert bins.shape[-1] == n_bins + 1
    eps = torch.finfo(weights.dtype).eps
    pdf = torch.nn.functional.normalize(weights, p=1, dim=-1)
    zero, one = torch.zeros_like(pdf[..., :1]), torch.ones_like(pdf[..., :1])
    cdf = torch.cat([zero, torch.cumsum(pdf[..., :-1], dim=-1), one], dim=-1)

    S = num_samples
    if stratified:
        u_max = eps + (1 - eps) / S
        max_jitter = (1 - u_max) / (S - 1) - eps
        u = torch.linspace(0, 1 - u_max, S, dtype=bins.dtype, device=bins.device)
        u = u + torch.rand(*bins.shape[:-1], 1, dtype=bins.dtype, device=bins.device) * max_jitter
    else:
        half = 1 / (2 * S)
        u = torch.linspace(half, 1 - half - eps, S, dtype=bins.dtype, device=bins.device)
        u = u.broadcast_to(bins.shape[:-1] + (S,))

    hi = torch.searchsorted(cdf.contiguous(), u.contiguous(), side="right")
    lo = hi - 1
    c_lo, c_hi = cdf.gather(-1, lo), cdf.gather(-1, hi)
    b_lo, b_hi = bins.gather(-1, lo), bins.gather(-1, hi)
    frac = (u - c_lo) / torch.clamp(c_hi - c_lo, min=eps)
    centres = b_lo + frac * (b_hi - b_lo)

    mids = (centres[..., 1:] + centres[..., :-1]) / 2
    first = (2 * centres[..., :1] - mids[..., :1]).clamp_min(vmin)
    last = (2 * centres[..., -1:] - mids[..., -1:]).clamp_max(vmax)
    return torch.cat([first, mids, last], dim=-1), centres
