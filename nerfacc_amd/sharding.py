"""Ray sharding across the GPUs of one node — new in this
implementation; the reference is single-GPU and has no distributed code at all.

Rays are independent, so the sampling / rendering kernels never communicate.  One process per
GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests);
every rank holds a full replica of the radiance field and of the OccGridEstimator, draws its
own 1/world of the ray batch, and the only exchange per step is

  * the all-reduce (average) of the radiance-field gradients: one flat fp32 buffer (the NGP field is
    ~12 M parameters = 49 MB), either as ONE message (`allreduce_gradients`) or — `ExchangeAdam` — cut
    into a few multi-MB chunks that are reduced asynchronously while the optimizer already updates the
    chunks that have arrived (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: chunks stay large
    enough to run at link bandwidth, few enough that launch latency does not add up), and
  * one all-reduce (sum) of two int64 scalars — rendered samples and rays of the step — so that
    every rank derives the same next ray-batch size (train_ngp_nerf_occ.py:187-194).

The occupancy grid stays identical on all ranks either by seeding its update identically
(`synchronized_rng`) or by `broadcast_grid` from rank 0 (2 MiB + 8 MiB at 128^3).
"""
from contextlib import contextmanager
from typing import Iterable, List, Tuple

import weakref

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
    _require_no_open_exchange("allreduce_gradients")
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
    _require_no_open_exchange("allreduce_counts")
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
    _require_no_open_exchange("allreduce_counts_begin")
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


# ExchangeAdam instances whose gradient exchange is in flight (first chunk launched from a backward hook, step() not yet
# run).  While one is, THIS module's other collectives refuse to start: a rank whose backward did not run (no samples)
# launches the same chunks later, from step(), so any collective issued in between would reach the process group in a
# different order on different ranks — a hang or a size mismatch.  Collectives the caller issues through
# torch.distributed directly cannot be checked: the rule is in ExchangeAdam's docstring.
_OPEN_EXCHANGES = weakref.WeakSet()          # optimizers with an exchange in flight (a WeakSet: a collected optimizer cannot leave its id behind)


def _require_no_open_exchange(what: str) -> None:
    if _OPEN_EXCHANGES:
        raise RuntimeError(f"nerfacc_amd.sharding.{what}: a gradient exchange started inside backward() is still in flight — no other "
                           "collective may be issued between backward() and ExchangeAdam.step() (ranks without samples launch the "
                           "same chunks from step(), so the order of collectives would differ between ranks); call it before "
                           "backward() or after step(), or build ExchangeAdam with overlap_backward=False")


class ExchangeAdam:
    """Adam over ONE flat fp32 parameter buffer with the gradient exchange folded in: the flat gradient is cut
    into `n_chunks` contiguous chunks, every chunk is exchanged asynchronously, and the Adam update of chunk k
    runs as soon as chunk k has arrived — while the other chunks are still on the wire.  Same update rule as
    torch.optim.Adam (L2 weight decay added to the gradient, bias correction, eps outside the square root).

    `params`: parameters whose storage is packed into the flat buffer (their .data / .grad become views of
    it, so autograd accumulates straight into the exchange buffer: no staging copies).

    **mode** — `"allreduce"`: every chunk is all-reduced and every rank runs Adam on the whole chunk (N identical
    updates).  `"rs_ag"`: every chunk is REDUCE-SCATTERED (each rank receives the sum of its 1/N of the chunk), Adam
    runs on that 1/N only — the optimizer pass and its moments' traffic shrink N-fold — and the updated 1/N is
    ALL-GATHERED back into every rank's parameters, chunk k's all-gather overlapping chunk k-1's Adam.  Same bytes
    on the wire as a ring all-reduce (which is a reduce-scatter followed by an all-gather), same result: every
    element's sum is formed once and every rank applies the same fp32 update to it.  The moments of a rank are
    authoritative on its own 1/N only; `state_dict()` gathers them (a collective: call it on every rank).

    **When a chunk's exchange starts.**  Default (`overlap_backward=False`): in `step()`, all chunks at once, in a
    fixed order.  With `overlap_backward=True` every parameter carries a post-accumulate-grad hook: a chunk is
    launched from inside the backward pass as soon as every parameter that overlaps it has its gradient — in a
    FIXED order, last chunk first (autograd reaches the last parameters of a model first), so that all ranks issue
    the same sequence of collectives whatever their graphs look like; a chunk that becomes ready out of turn waits
    for its predecessors.  `step()` launches what is left (ranks whose backward did not run — no samples this step —
    or parameters the loss did not reach) in the same order.  Two rules come with it, both enforced where this module
    can see the violation:
      * NO other collective between `backward()` and `step()` on any rank (a rank without samples launches its chunks
        from `step()`: anything in between would be ordered differently on different ranks).  This module's own
        collectives (`allreduce_counts[_begin]`, `broadcast_grid`, `allreduce_gradients`) raise while an exchange is
        in flight; collectives issued through torch.distributed directly are the caller's responsibility.
      * ONE backward per step: a second backward() would add into a buffer whose exchange is already in flight
        (silently wrong gradients) — the hook raises.  Accumulate gradients with `overlap_backward=False`.
    A field that is ONE tensor (bench.py's voxel grid) gets all its chunks launched by that tensor's hook, i.e. at
    the end of its backward kernel: nothing of the backward pass is left to overlap with, only the host time up
    to `optimizer.step()` and the Adam passes of the earlier chunks.

    The exchange runs whenever a process group is initialised — also with one rank (an RCCL all-reduce over a
    world of 1 is a valid collective: the single-GPU smoke test of this path); without a process group it is a
    plain chunked Adam.  The world size is read at construction (the flat buffers are padded so that every chunk
    splits evenly over the ranks).

    `timing = True` records HIP events around the exchange (`comm_stats()`): `wait_ms` — how long the compute
    stream stood still waiting for chunks (the exposed communication), `window_ms` — first launch to last arrival.
    `collective_log` (a list, when `record_collectives=True`) receives one `(op, chunk, numel)` tuple per collective
    issued, in issue order — what the multi-rank tests compare across ranks.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-2, betas=(0.9, 0.999), eps=1e-15,
                 weight_decay=0.0, n_chunks: int = 4, average: bool = True, overlap_backward: bool = False,
                 mode: str = "allreduce", record_collectives: bool = False):
        assert mode in ("allreduce", "rs_ag"), mode
        self.mode = mode
        self.params = [p for p in params if p.requires_grad]
        assert self.params and all(p.dtype == torch.float32 for p in self.params)
        dev = self.params[0].device
        self.ws = dist.get_world_size() if self._exchanging() else 1
        self.rank = dist.get_rank() if self._exchanging() else 0
        total = sum(p.numel() for p in self.params)
        unit = 256 * self.ws                                 # every chunk splits evenly over the ranks, shards start 1 KiB aligned
        padded = -(-total // unit) * unit
        self.total, self.padded = total, padded
        self.flat = torch.zeros(padded, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(padded, dtype=torch.float32, device=dev)
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            k = p.numel()
            self.offsets.append(off)
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            p.grad = self.grad[off:off + k].view_as(p.data)
            off += k
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.lr, self.betas, self.eps, self.weight_decay, self.average = lr, betas, eps, weight_decay, average
        self.t = 0
        n_chunks = max(1, min(int(n_chunks), padded // unit))
        size = -(-padded // n_chunks)
        size = -(-size // unit) * unit
        self.bounds = [(a, min(a + size, padded)) for a in range(0, padded, size)]
        # rs_ag: this rank's 1/N of chunk k is [a + rank * s, a + (rank + 1) * s), s = (b - a) / N; the reduced gradients
        # of all its shards live back to back in `gshard`
        self._shard = [((b - a) // self.ws) for a, b in self.bounds]
        self._gshard_off = [sum(self._shard[:k]) for k in range(len(self.bounds))]
        self.gshard = torch.zeros(sum(self._shard), dtype=torch.float32, device=dev) if mode == "rs_ag" else None
        self.step_tensor = torch.zeros((), dtype=torch.float32, device=dev)
        self._fused = dev.type == "cuda" and hasattr(torch, "_fused_adam_")
        # chunk k is complete when `_need[k]` parameters have reported; parameter i reports to `_chunks_of[i]`
        self._chunks_of: List[List[int]] = []
        self._need = [0] * len(self.bounds)
        for off, p in zip(self.offsets, self.params):
            ks = [k for k, (a, b) in enumerate(self.bounds) if off < b and off + p.numel() > a]
            self._chunks_of.append(ks)
            for k in ks:
                self._need[k] += 1
        self._have = [0] * len(self.bounds)
        self._reported = [False] * len(self.params)
        self._works: List = [None] * len(self.bounds)
        self._next = len(self.bounds) - 1                    # chunks are launched last to first
        self.overlap_backward = bool(overlap_backward) and hasattr(torch.Tensor, "register_post_accumulate_grad_hook")
        self._hooks = []
        if self.overlap_backward:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.timing = False
        self._events: List = []
        self._first_launch = None
        self.exchange_bytes = 4 * padded
        self.collective_log = [] if record_collectives else None

    # ---- bookkeeping ----------------------------------------------------------------------------------------------
    @staticmethod
    def _exchanging() -> bool:
        return dist.is_available() and dist.is_initialized()

    def _slot(self, i: int) -> torch.Tensor:
        p, off = self.params[i], self.offsets[i]
        return self.grad[off:off + p.numel()].view_as(p.data)

    def _bound(self, i: int) -> bool:
        g = self.params[i].grad
        return g is not None and g.data_ptr() == self.grad.data_ptr() + 4 * self.offsets[i] and g.is_contiguous()

    def _rebind(self, i: int) -> None:
        """p.grad was set to None (zero_grad(set_to_none=True)) or replaced: autograd then allocates gradients outside
        the flat buffer.  Move what is there into the parameter's slot and make p.grad the view again."""
        p, slot = self.params[i], self._slot(i)
        if p.grad is None:
            slot.zero_()
        else:
            slot.copy_(p.grad.to(torch.float32).reshape(slot.shape))
        p.grad = slot

    def zero_grad(self, set_to_none: bool = False) -> None:
        """zeroes the flat gradient; the parameters' .grad stay views of it whatever `set_to_none` says"""
        self.grad.zero_()
        for i in range(len(self.params)):
            if not self._bound(i):
                self.params[i].grad = self._slot(i)

    def _make_hook(self, i: int):
        def hook(_param):
            if not self._exchanging():
                return
            if self._reported[i]:
                if any(self._works[k] is not None for k in self._chunks_of[i]):
                    raise RuntimeError("ExchangeAdam(overlap_backward=True): a second backward() reached a parameter whose gradient "
                                       "chunk is already being exchanged — its contribution would be lost or double-counted.  Call "
                                       "step() after every backward(), or accumulate gradients with overlap_backward=False.")
                return
            if not self._bound(i):
                self._rebind(i)
            self._report(i)
            self._launch_ready()
        return hook

    def _report(self, i: int) -> None:
        self._reported[i] = True
        for k in self._chunks_of[i]:
            self._have[k] += 1

    def _log(self, op: str, k: int, numel: int) -> None:
        if self.collective_log is not None:
            self.collective_log.append((op, k, int(numel)))

    def _launch(self, k: int) -> None:
        a, b = self.bounds[k]
        if self.timing and self._first_launch is None and self.grad.is_cuda:
            self._first_launch = torch.cuda.Event(enable_timing=True)
            self._first_launch.record()
        if self.mode == "rs_ag":
            s, g0 = self._shard[k], self._gshard_off[k]
            self._works[k] = dist.reduce_scatter_tensor(self.gshard[g0:g0 + s], self.grad[a:b], op=dist.ReduceOp.SUM, async_op=True)
            self._log("reduce_scatter", k, b - a)
        else:
            self._works[k] = dist.all_reduce(self.grad[a:b], op=dist.ReduceOp.SUM, async_op=True)
            self._log("all_reduce", k, b - a)
        _OPEN_EXCHANGES.add(self)

    def _launch_ready(self) -> None:
        while self._next >= 0 and self._have[self._next] >= self._need[self._next]:
            self._launch(self._next)
            self._next -= 1

    # ---- the step -------------------------------------------------------------------------------------------------
    def _adam(self, p, g, m, v) -> None:
        b1, b2 = self.betas
        if self._fused:
            # one launch per chunk; state_steps holds the 1-based number of THIS step (torch's _fused_adam increments
            # its step tensors before the call), and the kernel does not modify it
            step = self.step_tensor
            torch._fused_adam_([p], [g], [m], [v], [], [step], lr=self.lr, beta1=b1, beta2=b2,
                               weight_decay=self.weight_decay, eps=self.eps, amsgrad=False, maximize=False)
            return
        if self.weight_decay:
            g = g.add(p, alpha=self.weight_decay)
        m.lerp_(g, 1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1, bc2 = 1.0 - b1**self.t, 1.0 - b2**self.t
        denom = (v.sqrt() / (bc2**0.5)).add_(self.eps)
        p.addcdiv_(m, denom, value=-self.lr / bc1)

    def _own(self, k: int):
        """[begin, end) of this rank's 1/N of chunk k in the flat buffers"""
        a, s = self.bounds[k][0], self._shard[k]
        return a + self.rank * s, a + (self.rank + 1) * s

    def step(self) -> None:
        """finish the exchange (launch the chunks the backward pass did not, in the fixed order) and update every chunk
        as it arrives, in arrival (= launch) order"""
        try:
            self._step()
        finally:
            # whatever happened (a world-size assert, a failed collective, a replaced gradient): the exchange is over as far as the
            # module's other collectives are concerned, and the next backward() starts from a clean slate.  Handles that
            # are still in flight — _step() raised between launching a chunk and waiting for it — are waited for first: RCCL is using
            # self.grad until then, and the next backward would write into it
            for w in self._works:
                if w is None:
                    continue
                for h in (w if isinstance(w, (list, tuple)) else (w,)):
                    try:
                        if h is not None and hasattr(h, "wait"):
                            h.wait()
                    except Exception:      # noqa: BLE001  (the exception that brought us here is the one to report)
                        pass
            self._first_launch = None
            self._works = [None] * len(self.bounds)
            self._have = [0] * len(self.bounds)
            self._reported = [False] * len(self.params)
            self._next = len(self.bounds) - 1
            _OPEN_EXCHANGES.discard(self)

    def _step(self) -> None:
        self.t += 1
        self.step_tensor += 1
        exchanging = self._exchanging()
        ws = dist.get_world_size() if exchanging else 1
        assert ws == self.ws, "ExchangeAdam: the world size changed after construction"
        for i in range(len(self.params)):                    # gradients that were produced outside the flat buffer
            if not self._bound(i):
                assert not self._reported[i], "ExchangeAdam: a gradient was replaced after its chunk had been sent"
                self._rebind(i)
        if exchanging:
            for i in range(len(self.params)):
                if not self._reported[i]:
                    self._report(i)
            self._launch_ready()
            assert self._next < 0
        timed = self.timing and self.grad.is_cuda and exchanging
        waits = []
        gathers = []
        for k in range(len(self.bounds) - 1, -1, -1):
            a, b = self.bounds[k]
            if exchanging:
                if timed:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                self._works[k].wait()                        # the current stream waits for this chunk only
                if timed:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    waits.append((e0, e1))
            if exchanging and self.mode == "rs_ag":
                s, g0 = self._shard[k], self._gshard_off[k]
                lo, hi = self._own(k)
                g = self.gshard[g0:g0 + s]
                if self.average and ws > 1:
                    g.div_(ws)
                self._adam(self.flat[lo:hi], g, self.m[lo:hi], self.v[lo:hi])
                # in place: the input is this rank's slice of the output (NCCL's in-place all-gather layout)
                gathers.append(dist.all_gather_into_tensor(self.flat[a:b], self.flat[lo:hi], async_op=True))
                self._log("all_gather", k, b - a)
            else:
                if exchanging and self.average and ws > 1:
                    self.grad[a:b].div_(ws)
                self._adam(self.flat[a:b], self.grad[a:b], self.m[a:b], self.v[a:b])
        for w in gathers:                                    # parameters are complete before anything reads them
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            w.wait()
            if timed:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                waits.append((e0, e1))
        if timed:
            self._events.append((self._first_launch, waits))
        self._first_launch = None
        self._works = [None] * len(self.bounds)
        self._have = [0] * len(self.bounds)
        self._reported = [False] * len(self.params)
        self._next = len(self.bounds) - 1
        _OPEN_EXCHANGES.discard(self)

    def set_mode(self, mode: str) -> None:
        """switch between "allreduce" and "rs_ag" between steps (a COLLECTIVE when leaving rs_ag: the moments are gathered so
        that every rank holds them whole again).  bench.py times both modes in one run with it."""
        assert mode in ("allreduce", "rs_ag"), mode
        assert all(w is None for w in self._works), "ExchangeAdam.set_mode: an exchange is in flight"
        if mode == self.mode:
            return
        if self.mode == "rs_ag":
            self._gather_moments()
        elif self.gshard is None:
            self.gshard = torch.zeros(sum(self._shard), dtype=torch.float32, device=self.flat.device)
        self.mode = mode

    def close(self) -> None:
        """remove the gradient hooks (the parameters stay views of the flat buffers)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.overlap_backward = False
        _OPEN_EXCHANGES.discard(self)

    def comm_stats(self, reset: bool = True) -> dict:
        """per-step averages over the steps taken with `timing = True` (synchronises the device)"""
        if not self._events:
            return {"steps": 0, "wait_ms": 0.0, "window_ms": 0.0, "exchange_bytes": self.exchange_bytes, "mode": self.mode}
        torch.cuda.synchronize(self.grad.device)
        wait = window = 0.0
        for first, waits in self._events:
            wait += sum(e0.elapsed_time(e1) for e0, e1 in waits)
            window += first.elapsed_time(waits[-1][1]) if first is not None else 0.0
        n = len(self._events)
        if reset:
            self._events = []
        return {"steps": n, "wait_ms": wait / n, "window_ms": window / n, "exchange_bytes": self.exchange_bytes, "mode": self.mode}

    # ---- checkpointing --------------------------------------------------------------------------------------------
    def _gather_moments(self) -> None:
        """rs_ag: a rank's moments are current on its own shards only — make them whole on every rank (collective)"""
        if self.mode != "rs_ag" or not self._exchanging() or self.ws == 1:
            return
        for buf in (self.m, self.v):
            for k, (a, b) in enumerate(self.bounds):
                lo, hi = self._own(k)
                dist.all_gather_into_tensor(buf[a:b], buf[lo:hi].clone())

    def state_dict(self) -> dict:
        """parameters, both moments, the step count.  In `rs_ag` mode this is a COLLECTIVE (the moments are gathered):
        call it on every rank."""
        self._gather_moments()
        n = self.total
        return {"flat": self.flat[:n].clone(), "m": self.m[:n].clone(), "v": self.v[:n].clone(), "t": self.t,
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay},
                "numels": [p.numel() for p in self.params]}

    def load_state_dict(self, state: dict) -> None:
        assert list(state["numels"]) == [p.numel() for p in self.params], "ExchangeAdam: parameter layout differs"
        n = self.total
        with torch.no_grad():
            self.flat[:n].copy_(state["flat"][:n])
            self.m[:n].copy_(state["m"][:n])
            self.v[:n].copy_(state["v"][:n])
        self.t = int(state["t"])
        self.step_tensor.fill_(float(self.t))
        h = state.get("hyper", {})
        self.lr, self.eps, self.weight_decay = h.get("lr", self.lr), h.get("eps", self.eps), h.get("weight_decay", self.weight_decay)
        self.betas = tuple(h.get("betas", self.betas))


def broadcast_grid(estimator, src: int = 0) -> None:
    """make `occs` and `binaries` of an OccGridEstimator identical to rank `src`'s."""
    rank, ws = world()
    if ws == 1:
        return
    _require_no_open_exchange("broadcast_grid")
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
