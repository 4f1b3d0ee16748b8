"""Ray sharding across the GPUs of one node — new in this
implementation; the reference is single-GPU and has no distributed code at all.

Rays are independent, so the sampling / rendering kernels never communicate.  One process per
GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests);
every rank holds a full replica of the radiance field and of the OccGridEstimator, draws its
own 1/world of the ray batch, and the only exchange per step is

  * ONE all-reduce (average) of the radiance-field gradients, issued as a single flat fp32
    bucket (the NGP field is ~12 M parameters = 49 MB; xGMI is point-to-point, 7 links x
    ~153 GB/s per GPU, so one large message beats many small ones), and
  * one all-reduce (sum) of two int64 scalars — rendered samples and rays of the step — so that
    every rank derives the same next ray-batch size (train_ngp_nerf_occ.py:187-194).

The occupancy grid stays identical on all ranks either by seeding its update identically
(`synchronized_rng`) or by `broadcast_grid` from rank 0 (2 MiB + 8 MiB at 128^3).
"""
from contextlib import contextmanager
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_rays: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[begin, end) of this rank's contiguous share of n_rays; shares differ by at most one ray."""
    base, extra = divmod(n_rays, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def allreduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = True) -> None:
    """One flat-bucket all-reduce over every parameter gradient (missing grads count as zero)."""
    rank, ws = world()
    if ws == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    grads: List[torch.Tensor] = []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    if len(grads) == 1 and grads[0].dtype == torch.float32 and grads[0].is_contiguous():
        # a single tensor already is the flat bucket: reduce it in place, no staging copies
        dist.all_reduce(grads[0], op=dist.ReduceOp.SUM)
        if average:
            grads[0].div_(ws)
        return
    flat = torch.cat([g.reshape(-1).to(torch.float32) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.div_(ws)
    off = 0
    for g in grads:
        k = g.numel()
        g.copy_(flat[off:off + k].view_as(g))
        off += k


def allreduce_counts(n_samples: int, n_rays: int, device) -> Tuple[int, int]:
    """global (samples, rays) of this step: one 16-byte all-reduce."""
    rank, ws = world()
    if ws == 1:
        return int(n_samples), int(n_rays)
    buf = torch.tensor([int(n_samples), int(n_rays)], dtype=torch.int64, device=device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    s, r = buf.tolist()
    return s, r


class _PendingCounts:
    __slots__ = ("buf", "work", "local")

    def __init__(self, buf, work, local):
        self.buf, self.work, self.local = buf, work, local


def allreduce_counts_begin(n_samples: int, n_rays: int, device) -> "_PendingCounts":
    """start the 16-byte all-reduce of this step's (samples, rays) without waiting for it.  Pair with
    `allreduce_counts_end` at the start of the next step: the host then blocks once — where it would
    wait for the previous step's GPU work anyway — instead of once more in the middle of the step."""
    rank, ws = world()
    if ws == 1:
        return _PendingCounts(None, None, (int(n_samples), int(n_rays)))
    buf = torch.tensor([int(n_samples), int(n_rays)], dtype=torch.int64, device=device)
    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
    return _PendingCounts(buf, work, None)


def allreduce_counts_end(pending: "_PendingCounts") -> Tuple[int, int]:
    """global (samples, rays) of the step `pending` was started in."""
    if pending.work is None:
        return pending.local
    pending.work.wait()              # the CURRENT stream waits for the collective (it may be a side stream)
    if pending.buf.is_cuda:
        pending.buf.record_stream(torch.cuda.current_stream(pending.buf.device))
    s, r = pending.buf.tolist()
    return s, r


def broadcast_grid(estimator, src: int = 0) -> None:
    """make `occs` and `binaries` of an OccGridEstimator identical to rank `src`'s."""
    rank, ws = world()
    if ws == 1:
        return
    dist.broadcast(estimator.occs, src=src)
    b = estimator.binaries.to(torch.uint8)
    dist.broadcast(b, src=src)
    estimator.binaries = b.to(torch.bool)
    if hasattr(estimator, "_occs_changed"):
        estimator._occs_changed()           # occs was overwritten in place: drop the memoised mean


@contextmanager
def synchronized_rng(seed: int, device=None):
    """Run a block with the SAME torch RNG stream on every rank (e.g. the occupancy-grid update),
    restoring each rank's own stream afterwards."""
    devices = []
    if device is not None and torch.device(device).type == "cuda":
        devices = [torch.device(device)]
    with torch.random.fork_rng(devices=devices):
        torch.manual_seed(seed)
        yield
