"""Per-ray inclusive / exclusive sum and product — nerfacc/scan.py.

Three input modes per function, as in the reference:
  * batched N-D tensor: plain torch.cumsum / cumprod along the last dim (differentiable);
  * flattened + `packed_info` [n_rays, 2]: chunked HIP scan (scan.hip, rows per quarter-wave);
  * flattened + `indices` (ray index per element): keyed HIP scan (ray-owning wave tiles).
Backward passes are the reversed scans the reference uses (scan.py:302-310, 400-404) and the
fused product backward of scan.cu:199-210.
"""
from typing import Optional

import torch
from torch import Tensor

from . import cuda as _C


def _mode(inputs: Tensor, packed_info: Optional[Tensor], indices: Optional[Tensor]) -> str:
    if indices is not None and packed_info is not None:
        raise ValueError("Only one of `indices` and `packed_info` can be specified.")
    if indices is not None:
        assert indices.dim() == 1 and indices.shape == inputs.shape, \
            "indices must be 1-D with the same shape as inputs."
        return "keyed"
    if packed_info is not None:
        assert inputs.dim() == 1, "inputs must be flattened."
        assert packed_info.dim() == 2 and packed_info.shape[-1] == 2, \
            "packed_info must be 2-D with shape (B, 2)."
        return "packed"
    return "batched"


def _shift(inputs: Tensor, fill: float) -> Tensor:
    pad = torch.full_like(inputs[..., :1], fill)
    return torch.cat([pad, inputs[..., :-1]], dim=-1)


def inclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None, indices: Optional[Tensor] = None) -> Tensor:
    """Running sum within each ray.

        >>> inclusive_sum(torch.arange(1., 10., device="cuda"), torch.tensor([[0, 2], [2, 3], [5, 4]], device="cuda"))
        tensor([ 1.,  3.,  3.,  7., 12.,  6., 13., 21., 30.], device='cuda:0')
    """
    mode = _mode(inputs, packed_info, indices)
    if mode == "keyed":
        return _KeyedSum.apply(indices, inputs, True)
    if mode == "packed":
        starts, cnts = packed_info.unbind(dim=-1)
        return _PackedSum.apply(starts, cnts, inputs, True, False)
    return torch.cumsum(inputs, dim=-1)


def exclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None, indices: Optional[Tensor] = None) -> Tensor:
    """Running sum within each ray, excluding the element itself (first element of a ray -> 0)."""
    mode = _mode(inputs, packed_info, indices)
    if mode == "keyed":
        return _KeyedSum.apply(indices, inputs, False)
    if mode == "packed":
        starts, cnts = packed_info.unbind(dim=-1)
        return _PackedSum.apply(starts, cnts, inputs, False, False)
    return torch.cumsum(_shift(inputs, 0.0), dim=-1)


def inclusive_prod(inputs: Tensor, packed_info: Optional[Tensor] = None, indices: Optional[Tensor] = None) -> Tensor:
    """Running product within each ray."""
    mode = _mode(inputs, packed_info, indices)
    if mode == "keyed":
        return _KeyedProd.apply(indices, inputs, True)
    if mode == "packed":
        starts, cnts = packed_info.unbind(dim=-1)
        return _PackedProd.apply(starts, cnts, inputs, True)
    return torch.cumprod(inputs, dim=-1)


def exclusive_prod(inputs: Tensor, packed_info: Optional[Tensor] = None, indices: Optional[Tensor] = None) -> Tensor:
    """Running product within each ray, excluding the element itself (first element -> 1)."""
    mode = _mode(inputs, packed_info, indices)
    if mode == "keyed":
        return _KeyedProd.apply(indices, inputs, False)
    if mode == "packed":
        starts, cnts = packed_info.unbind(dim=-1)
        return _PackedProd.apply(starts, cnts, inputs, False)
    return torch.cumprod(_shift(inputs, 1.0), dim=-1)


class _PackedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, starts, cnts, inputs, inclusive: bool, normalize: bool):
        starts, cnts, inputs = starts.contiguous(), cnts.contiguous(), inputs.contiguous()
        fn = _C.inclusive_sum if inclusive else _C.exclusive_sum
        out = fn(starts, cnts, inputs, normalize, False)
        if ctx.needs_input_grad[2]:
            ctx.inclusive, ctx.normalize = inclusive, normalize
            ctx.save_for_backward(starts, cnts)
        return out

    @staticmethod
    def backward(ctx, grad):
        assert not ctx.normalize, "Only support backward for normalize==False."
        starts, cnts = ctx.saved_tensors
        fn = _C.inclusive_sum if ctx.inclusive else _C.exclusive_sum
        return None, None, fn(starts, cnts, grad.contiguous(), False, True), None, None


class _PackedProd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, starts, cnts, inputs, inclusive: bool):
        starts, cnts, inputs = starts.contiguous(), cnts.contiguous(), inputs.contiguous()
        fn = _C.inclusive_prod_forward if inclusive else _C.exclusive_prod_forward
        out = fn(starts, cnts, inputs)
        if ctx.needs_input_grad[2]:
            ctx.inclusive = inclusive
            ctx.save_for_backward(starts, cnts, inputs, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        starts, cnts, inputs, out = ctx.saved_tensors
        fn = _C.inclusive_prod_backward if ctx.inclusive else _C.exclusive_prod_backward
        return None, None, fn(starts, cnts, inputs, out, grad.contiguous()), None


class _KeyedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, indices, inputs, inclusive: bool):
        indices, inputs = indices.contiguous(), inputs.contiguous()
        fn = _C.inclusive_sum_cub if inclusive else _C.exclusive_sum_cub
        out = fn(indices, inputs, False)
        if ctx.needs_input_grad[1]:
            ctx.inclusive = inclusive
            ctx.save_for_backward(indices)
        return out

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        fn = _C.inclusive_sum_cub if ctx.inclusive else _C.exclusive_sum_cub
        return None, fn(indices, grad.contiguous(), True), None


class _KeyedProd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, indices, inputs, inclusive: bool):
        indices, inputs = indices.contiguous(), inputs.contiguous()
        fn = _C.inclusive_prod_cub_forward if inclusive else _C.exclusive_prod_cub_forward
        out = fn(indices, inputs)
        if ctx.needs_input_grad[1]:
            ctx.inclusive = inclusive
            ctx.save_for_backward(indices, inputs, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        indices, inputs, out = ctx.saved_tensors
        fn = _C.inclusive_prod_cub_backward if ctx.inclusive else _C.exclusive_prod_cub_backward
        return None, fn(indices, inputs, out, grad.contiguous()), None
