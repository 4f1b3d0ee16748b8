"""PropNetEstimator — nerfacc/estimators/prop_net.py (Mip-NeRF 360 proposal sampling).

All tensors on this path are batched (n_rays, n); the native pieces are the inverse-cdf
resampler and the per-ray searchsorted of pdf.hip, everything else is differentiable torch.
"""
from typing import Callable, List, Optional, Tuple

try:
    from typing import Literal
except ImportError:  # pragma: no cover
    from typing_extensions import Literal

import torch
from torch import Tensor

from .. import cuda as _C
from ..data_specs import RayIntervals
from ..pdf import importance_sampling, searchsorted
from ..volrend import render_transmittance_from_density
from .base import AbstractEstimator


def _edge_cdfs(trans: Tensor) -> Tensor:
    """cdf at the n+1 interval edges from the transmittance at the n interval starts."""
    return 1.0 - torch.cat([trans, torch.zeros_like(trans[:, :1])], dim=-1)


class _EdgeCdfs(torch.autograd.Function):
    """cdfs at a level's n + 1 edges from its densities in one kernel (pdf.hip: edge_cdfs_*): render_transmittance_from_density +
    `_edge_cdfs` of the reference (prop_net.py:99-112), differentiable in sigmas"""

    @staticmethod
    def forward(ctx, t_edges, sigmas):
        t_edges, sigmas = t_edges.contiguous(), sigmas.contiguous()
        need = ctx.needs_input_grad[1]
        cdfs, trans = _C.edge_cdfs_fwd(t_edges, sigmas, need)
        if need:
            ctx.save_for_backward(t_edges, trans)
        return cdfs

    @staticmethod
    def backward(ctx, g_cdfs):
        t_edges, trans = ctx.saved_tensors
        return None, _C.edge_cdfs_bwd(t_edges, trans, g_cdfs.contiguous())


def _level_cdfs(t_vals: Tensor, sigmas: Tensor) -> Tensor:
    """cdf at the edges t_vals (n_rays, n + 1) of a proposal level with densities sigmas (n_rays, n)"""
    if sigmas.is_cuda and sigmas.dtype == torch.float32 and t_vals.dtype == torch.float32 and sigmas.dim() == 2 \
            and sigmas.shape[-1] >= 1 and not (torch.is_grad_enabled() and t_vals.requires_grad):
        return _EdgeCdfs.apply(t_vals, sigmas)
    trans, _ = render_transmittance_from_density(t_vals[..., :-1], t_vals[..., 1:], sigmas)
    return _edge_cdfs(trans)


class PropNetEstimator(AbstractEstimator):
    """Samples along rays from the cdfs predicted by a cascade of proposal networks.

    Args:
        optimizer / scheduler: optional, used by `update_every_n_steps` to train the proposal
            networks on the histogram loss.
    """

    def __init__(self, optimizer: Optional[torch.optim.Optimizer] = None,
                 scheduler: Optional[torch.optim.lr_scheduler._LRScheduler] = None) -> None:
        super().__init__()
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.prop_cache: List = []

    @torch.no_grad()
    def sampling(self, prop_sigma_fns: List[Callable], prop_samples: List[int], num_samples: int, n_rays: int,
                 near_plane: float, far_plane: float, sampling_type: Literal["uniform", "lindisp"] = "lindisp",
                 stratified: bool = False, requires_grad: bool = False) -> Tuple[Tensor, Tensor]:
        """Coarse-to-fine resampling (prop_net.py:37-129).  Each `prop_sigma_fns[k](t_starts,
        t_ends) -> sigmas` (n_rays, prop_samples[k]) refines the cdf from which the next level —
        and finally `num_samples` samples — are drawn.  Returns t_starts, t_ends of shape
        (n_rays, num_samples).  With `requires_grad` the proposal outputs are kept (with
        graph) for `compute_loss`."""
        assert len(prop_sigma_fns) == len(prop_samples), (
            "The number of proposal networks and the number of samples should be the same.")
        cdfs = torch.cat([torch.zeros((n_rays, 1), device=self.device), torch.ones((n_rays, 1), device=self.device)], dim=-1)
        intervals = RayIntervals(vals=cdfs)

        for level_fn, level_samples in zip(prop_sigma_fns, prop_samples):
            intervals, _ = importance_sampling(intervals, cdfs, level_samples, stratified)
            t_vals = _transform_stot(sampling_type, intervals.vals, near_plane, far_plane)
            t_starts, t_ends = t_vals[..., :-1], t_vals[..., 1:]
            with torch.set_grad_enabled(requires_grad):
                sigmas = level_fn(t_starts, t_ends)
                assert sigmas.shape == t_starts.shape
                cdfs = _level_cdfs(t_vals, sigmas)
                if requires_grad:
                    self.prop_cache.append((intervals, cdfs))

        intervals, _ = importance_sampling(intervals, cdfs, num_samples, stratified)
        t_vals = _transform_stot(sampling_type, intervals.vals, near_plane, far_plane)
        if requires_grad:
            self.prop_cache.append((intervals, None))
        return t_vals[..., :-1], t_vals[..., 1:]

    @torch.enable_grad()
    def compute_loss(self, trans: Tensor, loss_scaler: float = 1.0) -> Tensor:
        """Histogram-envelope loss of every cached proposal level against the final samples'
        transmittance `trans` (n_rays, num_samples) (prop_net.py:131-154)."""
        if len(self.prop_cache) == 0:
            return torch.zeros((), device=self.device)
        intervals, _ = self.prop_cache.pop()
        cdfs = _edge_cdfs(trans).detach()
        loss = 0.0
        while self.prop_cache:
            prop_intervals, prop_cdfs = self.prop_cache.pop()
            loss += _pdf_loss(intervals, cdfs, prop_intervals, prop_cdfs).mean()
        return loss * loss_scaler

    @torch.enable_grad()
    def update_every_n_steps(self, trans: Tensor, requires_grad: bool = False, loss_scaler: float = 1.0) -> float:
        """Step the proposal optimiser on the loss when `requires_grad`, else only the scheduler
        (prop_net.py:156-181).  Returns the loss value for logging."""
        if requires_grad:
            return self._update(trans=trans, loss_scaler=loss_scaler)
        if self.scheduler is not None:
            self.scheduler.step()
        return 0.0

    @torch.enable_grad()
    def _update(self, trans: Tensor, loss_scaler: float = 1.0) -> float:
        assert len(self.prop_cache) > 0
        assert self.optimizer is not None, "No optimizer is provided."
        loss = self.compute_loss(trans, loss_scaler)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return loss.item()


def get_proposal_requires_grad_fn(target: float = 5.0, num_steps: int = 1000) -> Callable:
    """Schedule deciding at which steps the proposal networks get gradients: the gap between
    such steps grows linearly to `target` over `num_steps` (prop_net.py:198-212)."""
    since_last = 0

    def proposal_requires_grad_fn(step: int) -> bool:
        nonlocal since_last
        wanted_gap = min(step / num_steps, 1.0) * target
        fire = since_last > wanted_gap
        if fire:
            since_last = 0
        since_last += 1
        return fire

    return proposal_requires_grad_fn


def _transform_stot(transform_type: Literal["uniform", "lindisp"], s_vals: torch.Tensor, t_min, t_max) -> torch.Tensor:
    """Map normalised s in [0,1] to ray distance t, linearly in t ("uniform") or in 1/t ("lindisp")."""
    if transform_type in ("uniform", "lindisp") and s_vals.is_cuda and s_vals.dtype == torch.float32 \
            and isinstance(t_min, (int, float)) and isinstance(t_max, (int, float)) \
            and not (torch.is_grad_enabled() and s_vals.requires_grad):
        # the same float operations in one launch (nfa_transform_stot) instead of five elementwise ones
        return _C.transform_stot(s_vals.contiguous(), float(t_min), float(t_max), transform_type == "lindisp")
    if transform_type == "uniform":
        return s_vals * t_max + (1 - s_vals) * t_min
    if transform_type == "lindisp":
        return 1 / (s_vals * (1 / t_max) + (1 - s_vals) * (1 / t_min))
    raise ValueError(f"Unknown transform_type: {transform_type}")


class _PdfLossBatched(torch.autograd.Function):
    """`_pdf_loss` on batched CUDA tensors in one kernel (pdf.hip: pdf_loss_*), differentiable in cdfs_key"""

    @staticmethod
    def forward(ctx, q_vals, cdfs_query, k_vals, cdfs_key, eps):
        need = ctx.needs_input_grad[3]
        loss, il, ir, coef = _C.pdf_loss_fwd(q_vals.contiguous(), cdfs_query.contiguous(), k_vals.contiguous(), cdfs_key.contiguous(),
                                             float(eps), need)
        if need:
            ctx.save_for_backward(il, ir, coef)
            ctx.n_key = k_vals.shape[-1] - 1
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        il, ir, coef = ctx.saved_tensors
        return None, None, None, _C.pdf_loss_bwd(g_loss.contiguous(), il, ir, coef, ctx.n_key), None


def _pdf_loss(segments_query: RayIntervals, cdfs_query: torch.Tensor, segments_key: RayIntervals,
              cdfs_key: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """max(0, w - w_outer)^2 / (w + eps): w = query interval mass, w_outer = mass of the key
    intervals that overlap it (prop_net.py:232-256)."""
    qv, kv = segments_query.vals, segments_key.vals
    if qv.dim() == 2 and kv.dim() == 2 and qv.is_cuda and all(t.dtype == torch.float32 for t in (qv, kv, cdfs_query, cdfs_key)) \
            and cdfs_query.shape == qv.shape and cdfs_key.shape == kv.shape and qv.shape[0] == kv.shape[0] \
            and qv.shape[1] >= 2 and kv.shape[1] >= 2 \
            and not (torch.is_grad_enabled() and (cdfs_query.requires_grad or qv.requires_grad or kv.requires_grad)):
        return _PdfLossBatched.apply(qv, cdfs_query, kv, cdfs_key, eps)
    ids_left, ids_right = searchsorted(segments_key, segments_query)
    if segments_query.vals.dim() > 1:
        w = cdfs_query[..., 1:] - cdfs_query[..., :-1]
        ids_left, ids_right = ids_left[..., :-1], ids_right[..., 1:]
    else:
        assert segments_query.is_left is not None and segments_query.is_right is not None
        w = cdfs_query[segments_query.is_right] - cdfs_query[segments_query.is_left]
        ids_left, ids_right = ids_left[segments_query.is_left], ids_right[segments_query.is_right]
    w_outer = cdfs_key.gather(-1, ids_right) - cdfs_key.gather(-1, ids_left)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + eps)


def _outer(t0_starts, t0_ends, t1_starts, t1_ends, y1) -> torch.Tensor:
    """Upper bound on the mass histogram (t1, y1) assigns to each interval of t0 (pure torch twin)."""
    cum = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    lo = torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1
    hi = torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right")
    lo = lo.clamp(0, y1.shape[-1] - 1)
    hi = hi.clamp(0, y1.shape[-1] - 1)
    return torch.take_along_dim(cum[..., 1:], hi, dim=-1) - torch.take_along_dim(cum[..., :-1], lo, dim=-1)


def _lossfun_outer(t: torch.Tensor, w: torch.Tensor, t_env: torch.Tensor, w_env: torch.Tensor):
    """Mip-NeRF 360 proposal loss in pure torch (twin of :func:`_pdf_loss`, prop_net.py:296-313)."""
    eps = torch.finfo(t.dtype).eps
    w_outer = _outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + eps)
