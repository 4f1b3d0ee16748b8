"""Ragged per-ray containers, API-compatible with nerfacc/data_specs.py:12-180.

`RaySamples` (sample centres) and `RayIntervals` (interval edges) hold either batched
``[n_rays, n]`` values or a flattened ``[total]`` tensor described by ``packed_info``
(``[n_rays, 2]`` = start, count per ray) and/or ``ray_indices``.  `_to_cpp` / `_from_cpp`
convert to and from the backend's `RaySegmentsSpec` exactly like the reference does with its
pybind class.
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import cuda as _C


def _stack_packed(spec) -> Optional[torch.Tensor]:
    if spec.chunk_starts is None or spec.chunk_cnts is None:
        return None
    return torch.stack([spec.chunk_starts, spec.chunk_cnts], dim=-1)


@dataclass
class RaySamples:
    """Samples along rays; batched or flattened (needs packed_info or ray_indices)."""

    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    is_valid: Optional[torch.Tensor] = None

    def _to_cpp(self):
        # NB: the reference reads a non-existent `self.chunk_cnts` here (data_specs.py:57)
        # and therefore raises; this version does what was intended.
        spec = _C.RaySegmentsSpec()
        spec.vals = self.vals.contiguous()
        if self.packed_info is not None:
            spec.chunk_starts = self.packed_info[:, 0].contiguous()
            spec.chunk_cnts = self.packed_info[:, 1].contiguous()
        if self.ray_indices is not None:
            spec.ray_indices = self.ray_indices.contiguous()
        return spec

    @classmethod
    def _from_cpp(cls, spec):
        return cls(vals=spec.vals, packed_info=_stack_packed(spec), ray_indices=spec.ray_indices,
                   is_valid=spec.is_valid)

    @property
    def device(self) -> torch.device:
        return self.vals.device


@dataclass
class RayIntervals:
    """Interval edges along rays.  In the flattened form `is_left[i]` / `is_right[i]` say
    whether edge i opens / closes an interval, which lets adjacent intervals share an edge and
    non-adjacent ones not (nerfacc/data_specs.py:96-140)."""

    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    is_left: Optional[torch.Tensor] = None
    is_right: Optional[torch.Tensor] = None

    def _to_cpp(self):
        spec = _C.RaySegmentsSpec()
        spec.vals = self.vals.contiguous()
        if self.packed_info is not None:
            spec.chunk_starts = self.packed_info[:, 0].contiguous()
            spec.chunk_cnts = self.packed_info[:, 1].contiguous()
        if self.ray_indices is not None:
            spec.ray_indices = self.ray_indices.contiguous()
        if self.is_left is not None:
            spec.is_left = self.is_left.contiguous()
        if self.is_right is not None:
            spec.is_right = self.is_right.contiguous()
        return spec

    @classmethod
    def _from_cpp(cls, spec):
        return cls(vals=spec.vals, packed_info=_stack_packed(spec), ray_indices=spec.ray_indices,
                   is_left=spec.is_left, is_right=spec.is_right)

    @property
    def device(self) -> torch.device:
        return self.vals.device
