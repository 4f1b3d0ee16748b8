"""pack_info — nerfacc/pack.py:10-49."""
from typing import Optional

import torch
from torch import Tensor

from . import cuda as _C


@torch.no_grad()
def pack_info(ray_indices: Tensor, n_rays: Optional[int] = None) -> Tensor:
    """`ray_indices` [n_samples] -> `packed_info` [n_rays, 2] = (start, count).

    The reference builds it from an atomic histogram plus a cumsum (any order of indices) and,
    like here, supports device tensors only.  Ascending indices — what every sampler emits —
    take one kernel here: each ray binary-searches its first and last sample (no atomics);
    any other order is detected on the device and handled as the reference does (integer
    histogram + exclusive sum), with the same result as `index_add_` + `cumsum`.

        >>> pack_info(torch.tensor([0, 0, 1, 1, 1, 2, 2, 2, 2], device="cuda"), n_rays=3)
        tensor([[0, 2], [2, 3], [5, 4]], device='cuda:0')
    """
    assert ray_indices.dim() == 1, "ray_indices must be a 1D tensor with shape (n_samples)."
    if not ray_indices.is_cuda:
        raise NotImplementedError("Only support cuda inputs.")
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() > 0 else 0
    return _C.pack_info(ray_indices.contiguous().long(), int(n_rays))
