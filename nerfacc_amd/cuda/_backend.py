"""ctypes binding of libnerfacc_hip.so — the module object that plays the role of the
reference's compiled extension ``nerfacc.csrc`` / ``nerfacc_cuda``
(nerfacc/cuda/_backend.py:51-86, nerfacc/cuda/csrc/nerfacc.cpp:126-163).

``_C`` exposes the same 21 names with the same argument order and meaning as the pybind
module, taking torch tensors; under each name it validates like the reference's CHECK_INPUT
(device + contiguity, utils_cuda.cuh:12-17), allocates the outputs from the torch caching
allocator and enqueues the HIP kernels on the current torch stream through the C ABI
declared in include/nerfacc_hip.h.  PyTorch is plumbing here (memory, streams); all
arithmetic is in the shared library.

There is deliberately NO fallback: if the library is missing or a tensor is not on a HIP
device the call raises.  Build with ``python -m nerfacc_amd.build`` (hipcc, a few seconds).
"""
from __future__ import annotations

import ctypes
import threading
import os
import weakref
from ctypes import c_float, c_int32, c_int64, c_void_p
from typing import List, Optional, Tuple

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("NERFACC_AMD_LIB") or os.path.join(_PKG, "libnerfacc_hip.so")   # (the override is for instrumented builds, tools/phase_cycles.py)

NFA_OP_SUM, NFA_OP_PROD = 0, 1


# struct layouts and prototypes come from include/nerfacc_hip.h itself (_cabi.parse_header): the header is the only
# place where an entry point's signature is written down
from ._cabi import HEADER_PATH, parse_header  # noqa: E402

_STRUCTS, _SIGNATURES = parse_header()
_TraverseArgs = _STRUCTS["nfa_traverse_args"]
_RaySegments = _STRUCTS["nfa_ray_segments"]

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class KernelTimer:
    """Optional HIP-event timer around single-kernel C-ABI calls (bench.py's live roofline
    measurement).  Events are recorded on the torch current stream, the stream the kernels are
    launched on; nothing is synchronised until `summary()`."""

    def __init__(self, names=("traverse_fill", "traverse_count", "rendering_fwd", "rendering_bwd", "visibility")):
        self.names = set(names)
        self.records = {}

    def bracket(self, name, fn, *args):
        if name not in self.names:
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        self.records.setdefault(name, []).append((e0, e1))
        return rc

    def summary(self):
        """{name: (launches, average milliseconds per launch)}"""
        torch.cuda.synchronize()
        if BACKEND == "ext":
            return {k: (int(n), float(ms)) for k, (n, ms) in _C.timing_summary().items()}
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1)) for k, v in self.records.items()}


_timer: Optional[KernelTimer] = None


def set_kernel_timer(timer: Optional[KernelTimer]) -> None:
    """bracket the named single-kernel calls with HIP events on their launch stream (ctypes backend: torch events
    around the call; torch-extension backend: hipEventRecord inside the extension)"""
    global _timer
    _timer = timer
    if BACKEND == "ext":
        _C.set_timing(None if timer is None else sorted(timer.names))


def _call(name, fn, *args):
    if _timer is None:
        return fn(*args)
    return _timer.bracket(name, fn, *args)


def _missing_entry(name: str):
    def stub(*_a):
        raise RuntimeError(f"nerfacc_amd: the library named by NERFACC_AMD_LIB ({LIB_PATH}) does not export {name}")
    return stub


def load_library() -> ctypes.CDLL:
    """dlopen the C-ABI library and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            _jit_build()          # like the reference's first-import JIT build (nerfacc/cuda/_backend.py:59-82)
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"nerfacc_amd: {LIB_PATH} not found and could not be built. Build the HIP library first: "
                "`python -m nerfacc_amd.build` (needs hipcc; there is no CPU fallback)."
            )
        lib = ctypes.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in _SIGNATURES.items():
            if os.environ.get("NERFACC_AMD_LIB") and not hasattr(lib, name):
                # an explicitly named variant build (A/B against an older source tree) may predate an entry point: it loads, says
                # so once, and a call of the missing entry point fails with its name
                missing.append(name)
                setattr(lib, name, _missing_entry(name))
                continue
            fn = getattr(lib, name)  # AttributeError => the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        if missing:
            import warnings

            warnings.warn(f"nerfacc_amd: {LIB_PATH} lacks {len(missing)} entry point(s) declared in include/nerfacc_hip.h "
                          f"({', '.join(missing[:6])}{' ...' if len(missing) > 6 else ''}): a stale or mismatched build", RuntimeWarning)
        _lib = lib
    return _lib


def _jit_build() -> None:
    """first use without built artefacts: compile them in-tree with hipcc, as the reference JIT-builds its
    extension on first import when no prebuilt one exists.  Silent no-op when no compiler is around."""
    import shutil

    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        return
    try:
        from ..build import build

        print("nerfacc_amd: building the HIP library and the torch extension for gfx950 (first use, ~1-2 min) ...", flush=True)
        build(force=False, verbose=False)
    except Exception as e:      # noqa: BLE001  (the caller raises ImportError with instructions)
        print(f"nerfacc_amd: build failed: {e}", flush=True)


# sync blocks of the single-launch forms (include/nerfacc_hip.h: NFA_SYNC_BYTES, zero before first use, left zero): one per
# (host thread, device, stream), like the extension's
_NFA_SYNC_BYTES = 16384
_tls = threading.local()
_LAST_SPR: dict = {}


def _sync_block(dev, stream) -> torch.Tensor:
    blocks = getattr(_tls, "sync_blocks", None)
    if blocks is None:
        blocks = _tls.sync_blocks = {}
    key = (dev.index, stream)
    if key not in blocks:
        blocks[key] = torch.zeros(_NFA_SYNC_BYTES, dtype=torch.uint8, device=dev)
    return blocks[key]


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("nerfacc_amd: " + load_library().nfa_last_error().decode())


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(ref: torch.Tensor):
    return torch.cuda.current_stream(ref.device).cuda_stream


def _check_input(t: torch.Tensor, name: str, dtype=None) -> None:
    # CHECK_INPUT of the reference (utils_cuda.cuh:12-17) + the dtype the kernels hard-code
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor (nerfacc_amd has no CPU kernels)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}, got {t.dtype}")


class _Guard:
    """OptionalCUDAGuard(device_of(t)) of the reference (utils_cuda.cuh:23-24)."""

    __slots__ = ("ctx",)

    def __init__(self, t: torch.Tensor):
        self.ctx = None
        if t.device.index is not None and t.device.index != torch.cuda.current_device():
            self.ctx = torch.cuda.device(t.device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


class _PyRaySegmentsSpec:
    """Mirror of the pybind class RaySegmentsSpec (nerfacc.cpp:128-137, data_spec.hpp:6-14):
    seven optional tensors with default construction and read/write attributes."""

    __slots__ = ("vals", "is_left", "is_right", "is_valid", "chunk_starts", "chunk_cnts", "ray_indices")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)

    def check(self) -> None:  # data_spec.hpp:15-51
        _check_input(self.vals, "vals", torch.float32)
        if self.vals.dim() > 1:
            return
        for k in ("chunk_starts", "chunk_cnts"):
            v = getattr(self, k)
            if v is None:
                raise RuntimeError(f"flattened RaySegmentsSpec needs {k}")
            _check_input(v, k, torch.int64)
            if v.dim() != 1:
                raise RuntimeError(f"{k} must be 1-D")
        if self.chunk_starts.numel() != self.chunk_cnts.numel():
            raise RuntimeError("chunk_starts and chunk_cnts differ in length")
        for k, dt in (("ray_indices", torch.int64), ("is_left", torch.bool), ("is_right", torch.bool), ("is_valid", torch.bool)):
            v = getattr(self, k)
            if v is not None:
                _check_input(v, k, dt)
                if v.dim() != 1 or v.numel() != self.vals.numel():
                    raise RuntimeError(f"{k} must be 1-D with as many elements as vals")

    def _view(self) -> _RaySegments:
        s = _RaySegments()
        s.vals = _ptr(self.vals)
        s.n_edges = self.vals.numel()
        if self.vals.dim() > 1:
            s.n_edges_per_ray = self.vals.shape[-1]
            s.n_rays = self.vals.numel() // max(self.vals.shape[-1], 1)
        else:
            s.chunk_starts = _ptr(self.chunk_starts)
            s.chunk_cnts = _ptr(self.chunk_cnts)
            s.ray_indices = _ptr(self.ray_indices)
            s.n_rays = self.chunk_cnts.numel()
        return s


# --------------------------------------------------------------------------------------
# host readback of the few integers a call needs (sample totals, kept counts).  The kernels
# store them straight into pinned host memory (device-visible on ROCm), so the readback is a
# stream wait plus a CPU load — no D2H copy to launch.
# --------------------------------------------------------------------------------------
_pinned = {}
_pinned_lock = threading.Lock()


def _host_ints(device: torch.device, n: int = 4) -> torch.Tensor:
    """one pinned slot per (device, stream, host thread): a kernel writes it on the calling thread's current
    stream and the same thread reads it after synchronising that stream, so neither another host thread
    nor another stream of the same device (an evaluation stream next to a training stream) can overwrite
    the totals in between — the reference functions are stateless, this keeps the calls re-entrant"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, threading.get_ident(), n)
    buf = _pinned.get(key)
    if buf is None:
        buf = torch.zeros(n, dtype=torch.int64).pin_memory()
        with _pinned_lock:
            if len(_pinned) > 256:          # streams / threads come and go: do not grow without bound
                _pinned.clear()
            _pinned[key] = buf
    return buf


def _read_ints(buf: torch.Tensor, device: torch.device):
    torch.cuda.current_stream(device).synchronize()
    return buf.tolist()


# --------------------------------------------------------------------------------------
# occupancy bricks: packed once per distinct `binaries` tensor state
# --------------------------------------------------------------------------------------
_BRICK_CACHE_SLOTS = 4
_brick_cache: List[dict] = []      # most recently used first: {"ref", "version", "bricks", "nonempty", "stream", "event"}
_brick_lock = threading.RLock()    # the cache is shared by every host thread


def _brick_entry(binaries: torch.Tensor) -> dict:
    """cache entry of `binaries` (packing it if its state is new).  A few slots, so that a training
    and an evaluation estimator — or a teacher and a student — used in turn do not repack (and
    re-read the non-empty count) on every call."""
    _check_input(binaries, "binaries", torch.bool)
    if binaries.dim() != 4:
        raise RuntimeError("binaries must have shape [n_grids, resx, resy, resz]")
    with _brick_lock:
        stream = torch.cuda.current_stream(binaries.device)
        for k, c in enumerate(_brick_cache):
            if c["ref"]() is binaries and c["version"] == binaries._version:
                if k:
                    _brick_cache.insert(0, _brick_cache.pop(k))
                if c["stream"] != stream.cuda_stream:
                    stream.wait_event(c["event"])      # packed on another stream: order this one after the pack
                return c
        L = load_library()
        G, rx, ry, rz = binaries.shape
        words = L.nfa_packed_grid_words(G, rx, ry, rz)
        bricks = torch.empty(words, dtype=torch.int64, device=binaries.device)
        with _Guard(binaries):
            _check(L.nfa_pack_binaries(_ptr(binaries), G, rx, ry, rz, _ptr(bricks), _stream(binaries)))
            return _insert_brick_entry(binaries, bricks)


def _insert_brick_entry(binaries: torch.Tensor, bricks: torch.Tensor) -> dict:
    """(caller holds _brick_lock and the device guard) make `bricks`, packed on the current stream, the front entry
    for this state of `binaries`"""
    stream = torch.cuda.current_stream(binaries.device)
    event = torch.cuda.Event()
    event.record(stream)
    # drop slots whose tensor is gone or has changed, then the oldest
    _brick_cache[:] = [c for c in _brick_cache if c["ref"]() is not None and not (c["ref"]() is binaries)]
    entry = {"ref": weakref.ref(binaries), "version": binaries._version, "bricks": bricks, "nonempty": -1,
             "stream": stream.cuda_stream, "event": event}
    _brick_cache.insert(0, entry)
    del _brick_cache[_BRICK_CACHE_SLOTS:]
    return entry


def packed_bricks(binaries: torch.Tensor) -> torch.Tensor:
    """bool [G, rx, ry, rz] -> uint64-as-int64 bricks (nfa_pack_binaries), memoised on the
    tensor object + its in-place version counter so that a grid that has not changed since
    the last call is not repacked (OccGridEstimator changes it every 16 steps)."""
    if BACKEND == "ext":
        return _C.packed_bricks(binaries)
    return _brick_entry(binaries)["bricks"]


def _nonempty_bricks(binaries: torch.Tensor) -> int:
    """number of non-empty bricks of the packed grid: one readback per grid state (every 16 training
    steps); lets the kernels size their LDS occupancy image to the grid instead of to the worst case"""
    c = _brick_entry(binaries)
    with _brick_lock:
        return _nonempty_locked(c, binaries)


def _nonempty_locked(c: dict, binaries: torch.Tensor) -> int:
    if c["nonempty"] < 0:
        G = binaries.shape[0]
        n_bricks_ = G * ((binaries.shape[1] + 3) // 4) * ((binaries.shape[2] + 3) // 4) * ((binaries.shape[3] + 3) // 4)
        hdr = c["bricks"][n_bricks_:n_bricks_ + 9].tolist()          # [0] non-empty bricks, [1..8] occupied voxels per level
        c["nonempty"], c["level_counts"] = int(hdr[0]), [int(x) for x in hdr[1:1 + min(G, 8)]]
    return c["nonempty"]


def _traverse_args(rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits,
                   near_planes, far_planes, step_size, cone_angle, limit, near_plane=0.0, far_plane=float("inf"),
                   t_min=None, t_max=None, jitter=None, jitter_scale=0.0) -> _TraverseArgs:
    _check_input(rays_o, "rays_o", torch.float32)
    _check_input(rays_d, "rays_d", torch.float32)
    _check_input(aabbs, "aabbs", torch.float32)
    n_rays = rays_o.shape[0]
    for t, nm in ((near_planes, "near_planes"), (far_planes, "far_planes"), (t_min, "t_min"), (t_max, "t_max"), (jitter, "jitter")):
        if t is not None:
            _check_input(t, nm, torch.float32)
            if t.numel() != n_rays:
                raise RuntimeError(f"{nm} must have n_rays elements")
    G = binaries.shape[0]
    if rays_o.shape != (n_rays, 3) or rays_d.shape != (n_rays, 3):
        raise RuntimeError("rays_o / rays_d must have shape [n_rays, 3]")
    if aabbs.shape != (G, 6):
        raise RuntimeError("aabbs must have shape [n_grids, 6]")
    a = _TraverseArgs()
    a.near_plane, a.far_plane, a.jitter_scale = near_plane, far_plane, jitter_scale
    a.t_min, a.t_max, a.jitter = _ptr(t_min), _ptr(t_max), _ptr(jitter)
    a.n_rays = n_rays
    a.rays_o, a.rays_d = _ptr(rays_o), _ptr(rays_d)
    if rays_mask is not None:
        _check_input(rays_mask, "rays_mask", torch.bool)
        a.rays_mask = _ptr(rays_mask)
    a.n_grids = G
    a.res[0], a.res[1], a.res[2] = binaries.shape[1], binaries.shape[2], binaries.shape[3]
    a.bricks = _ptr(packed_bricks(binaries))
    a.n_nonempty_bricks = _nonempty_bricks(binaries)
    a.aabbs = _ptr(aabbs)
    if t_sorted is not None:
        _check_input(t_sorted, "t_sorted", torch.float32)
        _check_input(t_indices, "t_indices", torch.int64)
        _check_input(hits, "hits", torch.bool)
        if t_sorted.shape != (n_rays, 2 * G) or t_indices.shape != (n_rays, 2 * G) or hits.shape != (n_rays, G):
            raise RuntimeError("t_sorted/t_indices must be [n_rays, 2*n_grids], hits [n_rays, n_grids]")
        a.t_sorted, a.t_indices, a.hits = _ptr(t_sorted), _ptr(t_indices), _ptr(hits)
    a.near_planes, a.far_planes = _ptr(near_planes), _ptr(far_planes)
    a.step_size, a.cone_angle, a.traverse_steps_limit = step_size, cone_angle, limit
    return a


def _rows(k: int, n: int, device) -> torch.Tensor:
    """k float rows of n elements from one allocation, each row 16-byte aligned (the tiled kernels then take
    2 or 4 elements per lane); rows are x[i] (1-D, contiguous)."""
    pitch = (n + 3) & ~3
    return torch.empty((k, pitch), dtype=torch.float32, device=device)[:, :n]


class _CtypesC:
    """ctypes face of the C ABI with the names of the reference's compiled module (the fallback backend:
    `NERFACC_AMD_BACKEND=ctypes`, or when the torch extension nerfacc_amd/_hip*.so is not built)."""

    RaySegmentsSpec = _PyRaySegmentsSpec

    # ---------------------------------------------------------------- misc
    @staticmethod
    def is_cub_available() -> bool:
        # scan-by-key is implemented natively (scan.hip), so the keyed path is always there
        # (reference: scan_cub.cu:58-64 returns whether CUB >= 1.15 was compiled in)
        load_library()
        return True

    # ---------------------------------------------------------------- grid
    @staticmethod
    def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane: float, far_plane: float, miss_value: float):
        """nerfacc.cpp:63-69 -> [t_mins, t_maxs, hits], each [n_rays, n_aabbs]."""
        for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (aabbs, "aabbs")):
            _check_input(t, n, torch.float32)
        R, G = rays_o.shape[0], aabbs.shape[0]
        t_mins = torch.empty((R, G), dtype=torch.float32, device=rays_o.device)
        t_maxs = torch.empty_like(t_mins)
        hits = torch.empty((R, G), dtype=torch.bool, device=rays_o.device)
        with _Guard(rays_o):
            _check(load_library().nfa_ray_aabb_intersect(
                _ptr(rays_o), _ptr(rays_d), R, _ptr(aabbs), G, near_plane, far_plane, miss_value,
                _ptr(t_mins), _ptr(t_maxs), _ptr(hits), _stream(rays_o)))
        return [t_mins, t_maxs, hits]

    @staticmethod
    def traverse_grids(rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits,
                       near_planes, far_planes, step_size: float, cone_angle: float,
                       compute_intervals: bool, compute_samples: bool, compute_terminate_planes: bool,
                       traverse_steps_limit: int, over_allocate: bool):
        """nerfacc.cpp:71-98 / grid.cu:320-474: returns (intervals, samples, terminate_planes).

        Two-pass mode: count -> offsets (device) -> ONE 16-byte readback -> allocate -> fill.
        As in the reference the two-pass mode ignores rays_mask (grid.cu:418,450)."""
        if over_allocate and traverse_steps_limit <= 0:
            raise RuntimeError("traverse_steps_limit must be > 0 when over_allocate is true")  # grid.cu:345
        L = load_library()
        _check_input(rays_o, "rays_o", torch.float32)
        dev = rays_o.device
        R = rays_o.shape[0]
        i64 = dict(dtype=torch.int64, device=dev)
        intervals, samples = _PyRaySegmentsSpec(), _PyRaySegmentsSpec()
        terminate = torch.empty((R,), dtype=torch.float32, device=dev) if compute_terminate_planes else None
        with _Guard(rays_o):
            stream = _stream(rays_o)
            a = _traverse_args(rays_o, rays_d, rays_mask if over_allocate else None, binaries, aabbs,
                               t_sorted, t_indices, hits, near_planes, far_planes, step_size, cone_angle,
                               int(traverse_steps_limit))
            if over_allocate:
                # grid.cu:364-404: fixed-size slots, single pass, then starts from actual counts
                maskl = rays_mask.to(torch.int64)
                iv_cnts = maskl * (2 * traverse_steps_limit)
                sm_cnts = maskl * traverse_steps_limit
                iv_starts, sm_starts = torch.empty_like(iv_cnts), torch.empty_like(sm_cnts)
                totals = torch.empty(2, **i64)
                _check(L.nfa_exclusive_sum_i64(_ptr(iv_cnts), R, _ptr(iv_starts), totals.data_ptr(), stream))
                _check(L.nfa_exclusive_sum_i64(_ptr(sm_cnts), R, _ptr(sm_starts), totals.data_ptr() + 8, stream))
                n_edges, n_samples = totals.tolist()
            else:
                iv_cnts = torch.empty(R, **i64) if compute_intervals else None
                iv_starts = torch.empty(R, **i64) if compute_intervals else None
                sm_cnts, sm_starts = torch.empty(R, **i64), torch.empty(R, **i64)
                totals = _host_ints(dev)
                a.workspace_bytes = L.nfa_traverse_workspace_bytes_for(ctypes.byref(a))
                ws = torch.empty(max(a.workspace_bytes, 16), dtype=torch.uint8, device=dev)
                a.iv_cnts, a.iv_starts = _ptr(iv_cnts), _ptr(iv_starts)
                a.sm_cnts, a.sm_starts, a.totals = _ptr(sm_cnts), _ptr(sm_starts), _ptr(totals)
                a.terminate_planes = _ptr(terminate)
                _check(L.nfa_traverse_count(ctypes.byref(a), _ptr(ws), stream))
                _check(L.nfa_traverse_offsets(ctypes.byref(a), _ptr(ws), stream))
                n_edges, n_samples, n_overflow, _ = _read_ints(totals, dev)   # the one host sync (data_spec.hpp:91)
            a.iv_cnts, a.iv_starts = _ptr(iv_cnts), _ptr(iv_starts)
            a.sm_cnts, a.sm_starts = _ptr(sm_cnts), _ptr(sm_starts)
            if compute_intervals:
                alloc = torch.zeros if over_allocate else torch.empty
                intervals.vals = alloc(n_edges, dtype=torch.float32, device=dev)
                intervals.ray_indices = alloc(n_edges, **i64)
                flags = torch.zeros((2, n_edges), dtype=torch.bool, device=dev)
                intervals.is_left, intervals.is_right = flags[0], flags[1]
                a.iv_vals, a.iv_ray_indices = _ptr(intervals.vals), _ptr(intervals.ray_indices)
                a.iv_is_left, a.iv_is_right = _ptr(intervals.is_left), _ptr(intervals.is_right)
            if compute_samples:
                alloc = torch.zeros if over_allocate else torch.empty
                samples.vals = alloc(n_samples, dtype=torch.float32, device=dev)
                samples.ray_indices = alloc(n_samples, **i64)
                samples.is_valid = (torch.zeros if over_allocate else torch.empty)(n_samples, dtype=torch.bool, device=dev)
                a.sm_vals, a.sm_ray_indices, a.sm_is_valid = _ptr(samples.vals), _ptr(samples.ray_indices), _ptr(samples.is_valid)
            a.terminate_planes = _ptr(terminate)
            if over_allocate:
                if R > 0:
                    _check(L.nfa_traverse_fill(ctypes.byref(a), 0, 1, None, 0, 0, stream))
            elif R > 0 and (compute_intervals or compute_samples) and n_samples > 0:
                _check(L.nfa_traverse_fill(ctypes.byref(a), 1, 0, _ptr(ws), n_samples, n_overflow, stream))
            if over_allocate:
                _check(L.nfa_exclusive_sum_i64(_ptr(iv_cnts), R, _ptr(iv_starts), None, stream))
                _check(L.nfa_exclusive_sum_i64(_ptr(sm_cnts), R, _ptr(sm_starts), None, stream))
        if compute_intervals:
            intervals.chunk_cnts, intervals.chunk_starts = iv_cnts, iv_starts
        if compute_samples:
            samples.chunk_cnts, samples.chunk_starts = sm_cnts, sm_starts
        return intervals, samples, terminate

    # ---------------------------------------------------------------- scans
    @staticmethod
    def _packed(chunk_starts, chunk_cnts, inputs, op, inclusive, reverse, normalize):
        _check_input(chunk_starts, "chunk_starts", torch.int64)
        _check_input(chunk_cnts, "chunk_cnts", torch.int64)
        _check_input(inputs, "inputs", torch.float32)
        if chunk_starts.dim() != 1 or chunk_cnts.dim() != 1 or inputs.dim() != 1:
            raise RuntimeError("chunk_starts, chunk_cnts and inputs must be 1-D")
        if chunk_starts.shape[0] != chunk_cnts.shape[0]:
            raise RuntimeError("chunk_starts and chunk_cnts differ in length")
        out = torch.empty_like(inputs)
        with _Guard(inputs):
            _check(load_library().nfa_scan_packed(_ptr(chunk_starts), _ptr(chunk_cnts), chunk_cnts.shape[0], _ptr(inputs),
                                                  _ptr(out), inputs.shape[0], op, int(inclusive), int(reverse),
                                                  int(normalize), _stream(inputs)))
        return out

    @staticmethod
    def _keyed(indices, inputs, op, inclusive, reverse):
        _check_input(indices, "indices", torch.int64)
        _check_input(inputs, "inputs", torch.float32)
        if indices.dim() != 1 or inputs.dim() != 1 or indices.shape[0] != inputs.shape[0]:
            raise RuntimeError("indices and inputs must be 1-D with the same length")
        out = torch.empty_like(inputs)
        with _Guard(inputs):
            _check(load_library().nfa_scan_keyed(_ptr(indices), _ptr(inputs), _ptr(out), inputs.shape[0], op,
                                                 int(inclusive), int(reverse), _stream(inputs)))
        return out

    @staticmethod
    def _prod_bwd(indices, chunk_starts, chunk_cnts, inputs, outputs, grad_outputs, inclusive):
        for t, n in ((inputs, "inputs"), (outputs, "outputs"), (grad_outputs, "grad_outputs")):
            _check_input(t, n, torch.float32)
        gin = torch.empty_like(grad_outputs)
        n_rays = 0 if chunk_cnts is None else chunk_cnts.shape[0]
        with _Guard(inputs):
            _check(load_library().nfa_prod_backward(_ptr(indices), _ptr(chunk_starts), _ptr(chunk_cnts), n_rays,
                                                    _ptr(inputs), _ptr(outputs), _ptr(grad_outputs), _ptr(gin),
                                                    inputs.shape[0], int(inclusive), _stream(inputs)))
        return gin

    @staticmethod
    def inclusive_sum(chunk_starts, chunk_cnts, inputs, normalize: bool, backward: bool):
        return _CtypesC._packed(chunk_starts, chunk_cnts, inputs, NFA_OP_SUM, True, backward, normalize)

    @staticmethod
    def exclusive_sum(chunk_starts, chunk_cnts, inputs, normalize: bool, backward: bool):
        return _CtypesC._packed(chunk_starts, chunk_cnts, inputs, NFA_OP_SUM, False, backward, normalize)

    @staticmethod
    def inclusive_prod_forward(chunk_starts, chunk_cnts, inputs):
        return _CtypesC._packed(chunk_starts, chunk_cnts, inputs, NFA_OP_PROD, True, False, False)

    @staticmethod
    def exclusive_prod_forward(chunk_starts, chunk_cnts, inputs):
        return _CtypesC._packed(chunk_starts, chunk_cnts, inputs, NFA_OP_PROD, False, False, False)

    @staticmethod
    def inclusive_prod_backward(chunk_starts, chunk_cnts, inputs, outputs, grad_outputs):
        _check_input(chunk_starts, "chunk_starts", torch.int64)
        _check_input(chunk_cnts, "chunk_cnts", torch.int64)
        return _CtypesC._prod_bwd(None, chunk_starts, chunk_cnts, inputs, outputs, grad_outputs, True)

    @staticmethod
    def exclusive_prod_backward(chunk_starts, chunk_cnts, inputs, outputs, grad_outputs):
        _check_input(chunk_starts, "chunk_starts", torch.int64)
        _check_input(chunk_cnts, "chunk_cnts", torch.int64)
        return _CtypesC._prod_bwd(None, chunk_starts, chunk_cnts, inputs, outputs, grad_outputs, False)

    @staticmethod
    def inclusive_sum_cub(indices, inputs, backward: bool):
        return _CtypesC._keyed(indices, inputs, NFA_OP_SUM, True, backward)

    @staticmethod
    def exclusive_sum_cub(indices, inputs, backward: bool):
        return _CtypesC._keyed(indices, inputs, NFA_OP_SUM, False, backward)

    @staticmethod
    def inclusive_prod_cub_forward(indices, inputs):
        return _CtypesC._keyed(indices, inputs, NFA_OP_PROD, True, False)

    @staticmethod
    def exclusive_prod_cub_forward(indices, inputs):
        return _CtypesC._keyed(indices, inputs, NFA_OP_PROD, False, False)

    @staticmethod
    def inclusive_prod_cub_backward(indices, inputs, outputs, grad_outputs):
        _check_input(indices, "indices", torch.int64)
        return _CtypesC._prod_bwd(indices, None, None, inputs, outputs, grad_outputs, True)

    @staticmethod
    def exclusive_prod_cub_backward(indices, inputs, outputs, grad_outputs):
        _check_input(indices, "indices", torch.int64)
        return _CtypesC._prod_bwd(indices, None, None, inputs, outputs, grad_outputs, False)

    # ---------------------------------------------------------------- pdf
    @staticmethod
    def importance_sampling(ray_segments: _PyRaySegmentsSpec, cdfs, n_intervels_per_ray, stratified: bool):
        """nerfacc.cpp:100-112: an int (batched outputs) or a per-ray Tensor of counts (flattened outputs; the reference's own
        implementation of that overload allocates zero elements, pdf.cu:324 — this one follows what its kernels state)."""
        ray_segments.check()
        _check_input(cdfs, "cdfs", torch.float32)
        if cdfs.numel() != ray_segments.vals.numel():
            raise RuntimeError("cdfs and ray_segments.vals must have the same number of elements")
        if isinstance(n_intervels_per_ray, torch.Tensor):
            view = ray_segments._view()
            if n_intervels_per_ray.dtype.is_floating_point or n_intervels_per_ray.dtype == torch.bool:
                raise RuntimeError(f"n_intervals_per_ray must be an integer tensor, got {n_intervels_per_ray.dtype}")
            cnts = n_intervels_per_ray.to(torch.int64).reshape(-1).contiguous()
            if cnts.device != cdfs.device or not cnts.is_cuda:
                raise RuntimeError("n_intervals_per_ray must live on the device of cdfs")
            if cnts.numel() != int(view.n_rays):
                raise RuntimeError(f"n_intervals_per_ray must hold one count per ray ({int(view.n_rays)}), got {cnts.numel()}")
            dev = cdfs.device
            samples, intervals = _PyRaySegmentsSpec(), _PyRaySegmentsSpec()
            samples.chunk_cnts = cnts
            cs = torch.cumsum(cnts, 0)
            samples.chunk_starts = cs - cnts
            intervals.chunk_cnts = (cnts + 1) * (cnts > 0).to(torch.int64)
            ics = torch.cumsum(intervals.chunk_cnts, 0)
            intervals.chunk_starts = ics - intervals.chunk_cnts
            n_s = n_e = 0
            if cnts.numel():            # one read-back: the smallest count and the two totals together
                lo, n_s, n_e = torch.stack([cnts.min(), cs[-1], ics[-1]]).tolist()
                if lo < 0:
                    raise RuntimeError("n_intervals_per_ray must not be negative")
            samples.vals = torch.empty(n_s, dtype=torch.float32, device=dev)
            samples.ray_indices = torch.empty(n_s, dtype=torch.int64, device=dev)
            intervals.vals = torch.empty(n_e, dtype=torch.float32, device=dev)
            intervals.ray_indices = torch.empty(n_e, dtype=torch.int64, device=dev)
            intervals.is_left = torch.empty(n_e, dtype=torch.bool, device=dev)
            intervals.is_right = torch.empty(n_e, dtype=torch.bool, device=dev)
            jitter = torch.rand(int(view.n_rays), dtype=torch.float32, device=dev) if stratified else None
            with _Guard(cdfs):
                _check(load_library().nfa_importance_sampling_ragged(
                    ctypes.byref(view), _ptr(cdfs), _ptr(samples.chunk_starts), _ptr(cnts), _ptr(intervals.chunk_starts), n_s, _ptr(jitter),
                    _ptr(samples.vals), _ptr(samples.ray_indices), _ptr(intervals.vals), _ptr(intervals.ray_indices),
                    _ptr(intervals.is_left), _ptr(intervals.is_right), _stream(cdfs)))
            return [intervals, samples]
        n = int(n_intervels_per_ray)
        view = ray_segments._view()
        if ray_segments.vals.dim() > 1:
            lead = list(ray_segments.vals.shape[:-1])
        else:
            lead = [int(view.n_rays)]
        dev = cdfs.device
        samples, intervals = _PyRaySegmentsSpec(), _PyRaySegmentsSpec()
        samples.vals = torch.empty(lead + [n], dtype=torch.float32, device=dev)
        intervals.vals = torch.empty(lead + [n + 1], dtype=torch.float32, device=dev)
        jitter = None
        if stratified:
            # one uniform per ray from torch's generator (the reference draws it with Philox
            # inside the kernel, pdf.cu:138-144: same distribution, different stream)
            jitter = torch.rand(int(view.n_rays), dtype=torch.float32, device=dev)
        with _Guard(cdfs):
            _check(load_library().nfa_importance_sampling(ctypes.byref(view), _ptr(cdfs), n, _ptr(jitter),
                                                          _ptr(intervals.vals), _ptr(samples.vals), _stream(cdfs)))
        return [intervals, samples]

    @staticmethod
    def searchsorted(query: _PyRaySegmentsSpec, key: _PyRaySegmentsSpec):
        """nerfacc.cpp:114-117 -> [ids_left, ids_right] shaped like query.vals."""
        query.check()
        key.check()
        ids_left = torch.empty(query.vals.shape, dtype=torch.int64, device=query.vals.device)
        ids_right = torch.empty_like(ids_left)
        q, k = query._view(), key._view()
        with _Guard(query.vals):
            _check(load_library().nfa_searchsorted(ctypes.byref(q), ctypes.byref(k), _ptr(ids_left), _ptr(ids_right),
                                                   _stream(query.vals)))
        return [ids_left, ids_right]

    @staticmethod
    def transform_stot(s_vals, t_min: float, t_max: float, lindisp: bool):
        """prop_net.py:215-229 in one launch (nfa_transform_stot)."""
        _check_input(s_vals, "s_vals", torch.float32)
        t = torch.empty_like(s_vals)
        with _Guard(s_vals):
            _check(load_library().nfa_transform_stot(_ptr(s_vals), s_vals.numel(), t_min, t_max, int(lindisp), _ptr(t), _stream(s_vals)))
        return t

    @staticmethod
    def edge_cdfs_fwd(t_edges, sigmas, want_trans: bool):
        """(cdfs [R, S + 1], trans [R, S] or None): nfa_edge_cdfs_fwd"""
        _check_input(t_edges, "t_edges", torch.float32)
        _check_input(sigmas, "sigmas", torch.float32)
        if not (sigmas.dim() == 2 and t_edges.dim() == 2 and t_edges.shape[0] == sigmas.shape[0] and t_edges.shape[1] == sigmas.shape[1] + 1
                and sigmas.shape[1] >= 1):
            raise RuntimeError("edge_cdfs: t_edges must be [n_rays, n + 1] and sigmas [n_rays, n]")
        cdfs = torch.empty_like(t_edges)
        trans = torch.empty_like(sigmas) if want_trans else None
        with _Guard(sigmas):
            _check(load_library().nfa_edge_cdfs_fwd(_ptr(t_edges), _ptr(sigmas), sigmas.shape[0], sigmas.shape[1], _ptr(cdfs), _ptr(trans),
                                                    _stream(sigmas)))
        return [cdfs, trans]

    @staticmethod
    def edge_cdfs_bwd(t_edges, trans, g_cdfs):
        for t, nm in ((t_edges, "t_edges"), (trans, "trans"), (g_cdfs, "g_cdfs")):
            _check_input(t, nm, torch.float32)
        g_sig = torch.empty_like(trans)
        with _Guard(trans):
            _check(load_library().nfa_edge_cdfs_bwd(_ptr(t_edges), _ptr(trans), _ptr(g_cdfs), trans.shape[0], trans.shape[1], _ptr(g_sig),
                                                    _stream(trans)))
        return g_sig

    @staticmethod
    def pdf_loss_fwd(q, cq, k, ck, eps: float, want_grad: bool):
        """(loss [R, Nq], ids_left, ids_right, coef): nfa_pdf_loss_fwd; the last three None unless want_grad"""
        for t, nm in ((q, "query vals"), (cq, "cdfs_query"), (k, "key vals"), (ck, "cdfs_key")):
            _check_input(t, nm, torch.float32)
        if not (q.dim() == 2 and k.dim() == 2 and cq.shape == q.shape and ck.shape == k.shape and q.shape[0] == k.shape[0]
                and q.shape[1] >= 2 and k.shape[1] >= 2):
            raise RuntimeError("pdf_loss: batched [n_rays, n + 1] edges and cdfs expected")
        R, nq, nk = q.shape[0], q.shape[1] - 1, k.shape[1] - 1
        loss = torch.empty((R, nq), dtype=torch.float32, device=q.device)
        il = ir = coef = None
        if want_grad:
            il = torch.empty((R, nq), dtype=torch.int32, device=q.device)
            ir, coef = torch.empty_like(il), torch.empty_like(loss)
        with _Guard(q):
            _check(load_library().nfa_pdf_loss_fwd(_ptr(q), _ptr(cq), _ptr(k), _ptr(ck), R, nq, nk, eps, _ptr(loss), _ptr(il), _ptr(ir),
                                                   _ptr(coef), _stream(q)))
        return [loss, il, ir, coef]

    @staticmethod
    def pdf_loss_bwd(g_loss, il, ir, coef, n_key: int):
        _check_input(g_loss, "g_loss", torch.float32)
        g_ck = torch.empty((g_loss.shape[0], n_key + 1), dtype=torch.float32, device=g_loss.device)
        with _Guard(g_loss):
            _check(load_library().nfa_pdf_loss_bwd(_ptr(g_loss), _ptr(il), _ptr(ir), _ptr(coef), g_loss.shape[0], g_loss.shape[1], n_key,
                                                   _ptr(g_ck), _stream(g_loss)))
        return g_ck

    # ---------------------------------------------------------------- camera (out of scope)
    @staticmethod
    def opencv_lens_undistortion(*args, **kwargs):
        raise NotImplementedError("camera undistortion (camera.cu) is outside the OccGrid hot path and not built")

    @staticmethod
    def opencv_lens_undistortion_fisheye(*args, **kwargs):
        raise NotImplementedError("camera undistortion (camera.cu) is outside the OccGrid hot path and not built")

    # ================================================================ fused entry points
    # (no counterpart in the reference's extension: there these are chains of ATen ops)
    @staticmethod
    def sample_occgrid(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size: float,
                       cone_angle: float, rays_mask=None, traverse_steps_limit: int = -1,
                       with_terminate_planes: bool = False, near_plane: float = 0.0, far_plane: float = float("inf"),
                       t_min=None, t_max=None, jitter=None, jitter_scale: float = 0.0):
        """traverse_grids + the two is_left/is_right compactions of occ_grid.py:164-177 in
        one count pass and one fill pass: returns (ray_indices, t_starts, t_ends, packed_info).
        Ray/AABB tests and the per-ray event sort run inside the kernel.

        rays_mask (bool [R], rays with False are skipped) and traverse_steps_limit (at most that
        many samples per ray) give one round of the test-time marcher (examples/utils.py:349-372)
        with exactly sized, already compacted outputs instead of over-allocation + masks;
        with_terminate_planes appends where each ray stopped (its near plane if it was skipped)."""
        L = load_library()
        _check_input(rays_o, "rays_o", torch.float32)
        dev = rays_o.device
        R = rays_o.shape[0]
        i64 = dict(dtype=torch.int64, device=dev)
        with _Guard(rays_o):
            stream = _stream(rays_o)
            a = _traverse_args(rays_o, rays_d, rays_mask, binaries, aabbs, None, None, None, near_planes, far_planes,
                               step_size, cone_angle, traverse_steps_limit, near_plane, far_plane, t_min, t_max, jitter,
                               jitter_scale)
            packed = torch.empty((2, R), **i64)          # [starts; cnts], stacked to [R,2] below
            totals = _host_ints(dev)
            a.workspace_bytes = L.nfa_traverse_workspace_bytes_for(ctypes.byref(a))
            ws = torch.empty(max(a.workspace_bytes, 16), dtype=torch.uint8, device=dev)
            a.sm_starts, a.sm_cnts, a.totals = packed[0].data_ptr(), packed[1].data_ptr(), _ptr(totals)
            term = None
            if with_terminate_planes:
                if near_planes is None:
                    raise RuntimeError("with_terminate_planes needs near_planes as a tensor")
                term = near_planes.clone()
                a.terminate_planes = _ptr(term)
            emitted, cap = False, 0
            if R > 0 and L.nfa_traverse_sample_fused(ctypes.byref(a)):
                # the whole call as ONE launch (round 6, include/nerfacc_hip.h: nfa_traverse_sample): outputs sized from this
                # device's previous call (samples per ray), offsets by look-back inside the count kernel, every wave expands its rays
                spr = _LAST_SPR.get(dev.index, 0.0)
                cap = int(1.25 * spr * R) + 1024 if spr > 0 else 0
                if cap:
                    ray_indices = torch.empty(cap, **i64)
                    ts = _rows(2, cap, dev)
                    a.sm_ray_indices, a.t_starts, a.t_ends = _ptr(ray_indices), ts[0].data_ptr(), ts[1].data_ptr()
                _check(_call("traverse_sample", L.nfa_traverse_sample, ctypes.byref(a), _ptr(ws), cap, 0, _ptr(_sync_block(dev, stream)),
                             None, stream))
                _, n, n_overflow, _ = _read_ints(totals, dev)
                emitted = cap > 0
                if n < 0:                                # its look-back gave up (bounded wait): offsets by their own kernel
                    _check(L.nfa_traverse_offsets(ctypes.byref(a), _ptr(ws), stream))
                    _, n, n_overflow, _ = _read_ints(totals, dev)
                    emitted = False
            else:
                _check(_call("traverse_count", L.nfa_traverse_count, ctypes.byref(a), _ptr(ws), stream))
                _check(L.nfa_traverse_offsets(ctypes.byref(a), _ptr(ws), stream))
                _, n, n_overflow, _ = _read_ints(totals, dev)
            if not (0 <= n and 0 <= n_overflow <= R):
                # (seen once in ~60 runs of eight processes sharing one GPU: the count launch lost one XCD's share of its workgroups'
                #  stores, profiles/r06_oversubscription.md — an error instead of outputs sized by garbage)
                raise RuntimeError(f"nerfacc_amd: sample_occgrid read back inconsistent totals (samples {n}, overflow rays {n_overflow} "
                                   f"of {R} rays): the count pass's outputs are corrupt")
            if R > 0:
                _LAST_SPR[dev.index] = n / R
            a.terminate_planes = None                    # written by the count pass only
            if emitted and n <= cap:
                if n_overflow > 0:
                    _check(L.nfa_traverse_fill(ctypes.byref(a), 1, 0, _ptr(ws), 0, n_overflow, stream))
                ray_indices, ts = ray_indices[:n], (ts[0][:n], ts[1][:n])
            else:
                ray_indices = torch.empty(n, **i64)
                ts = _rows(2, n, dev)
                a.sm_ray_indices, a.t_starts, a.t_ends = _ptr(ray_indices), ts[0].data_ptr(), ts[1].data_ptr()
                if n > 0:
                    _check(_call("traverse_fill", L.nfa_traverse_fill, ctypes.byref(a), 1, 0, _ptr(ws), n, n_overflow, stream))
        if with_terminate_planes:
            return ray_indices, ts[0], ts[1], packed.t(), term
        return ray_indices, ts[0], ts[1], packed.t()

    @staticmethod
    def pack_info(ray_indices, n_rays: int):
        _check_input(ray_indices, "ray_indices", torch.int64)
        out = torch.empty((n_rays, 2), dtype=torch.int64, device=ray_indices.device)
        with _Guard(ray_indices):
            _check(load_library().nfa_pack_info(_ptr(ray_indices), ray_indices.shape[0], n_rays, _ptr(out), _stream(ray_indices)))
        return out

    @staticmethod
    def unpack_info(chunk_starts, chunk_cnts, n: int):
        _check_input(chunk_starts, "chunk_starts", torch.int64)
        _check_input(chunk_cnts, "chunk_cnts", torch.int64)
        out = torch.empty(n, dtype=torch.int64, device=chunk_starts.device)
        with _Guard(chunk_starts):
            _check(load_library().nfa_unpack_info(_ptr(chunk_starts), _ptr(chunk_cnts), chunk_cnts.shape[0], _ptr(out), n,
                                                  _stream(chunk_starts)))
        return out

    @staticmethod
    def render_weight_from_density_fwd(ray_indices, t_starts, t_ends, sigmas, prefix_trans=None):
        _check_input(ray_indices, "ray_indices", torch.int64)
        for t, n in ((t_starts, "t_starts"), (t_ends, "t_ends"), (sigmas, "sigmas")):
            _check_input(t, n, torch.float32)
        if prefix_trans is not None:
            _check_input(prefix_trans, "prefix_trans", torch.float32)
        n = sigmas.shape[0]
        out = _rows(3, n, sigmas.device)
        with _Guard(sigmas):
            _check(load_library().nfa_render_weight_from_density_fwd(
                _ptr(ray_indices), _ptr(t_starts), _ptr(t_ends), _ptr(sigmas), _ptr(prefix_trans), n,
                out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _stream(sigmas)))
        return out[0], out[1], out[2]

    @staticmethod
    def render_weight_from_density_bwd(ray_indices, t_starts, t_ends, sigmas, trans, alphas, g_w, g_T, g_a):
        n = sigmas.shape[0]
        g = torch.empty_like(sigmas)
        for t, nm in ((g_w, "g_weights"), (g_T, "g_trans"), (g_a, "g_alphas")):
            if t is not None:
                _check_input(t, nm, torch.float32)
        with _Guard(sigmas):
            _check(load_library().nfa_render_weight_from_density_bwd(
                _ptr(ray_indices), _ptr(t_starts), _ptr(t_ends), _ptr(sigmas), _ptr(trans), _ptr(alphas),
                _ptr(g_w), _ptr(g_T), _ptr(g_a), n, _ptr(g), _stream(sigmas)))
        return g

    @staticmethod
    def sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends, with_dirs: bool = False):
        """positions [N,3] = rays_o[ray_indices] + rays_d[ray_indices] * ((t_starts + t_ends)[:, None] / 2)
        in one launch (bit-identical to that torch expression); with_dirs also returns rays_d[ray_indices]."""
        _check_input(rays_o, "rays_o", torch.float32)
        _check_input(rays_d, "rays_d", torch.float32)
        _check_input(ray_indices, "ray_indices", torch.int64)
        _check_input(t_starts, "t_starts", torch.float32)
        _check_input(t_ends, "t_ends", torch.float32)
        n = ray_indices.shape[0]
        if t_starts.shape[0] != n or t_ends.shape[0] != n or rays_o.shape != rays_d.shape or rays_o.dim() != 2 or rays_o.shape[1] != 3:
            raise RuntimeError("sample_positions: rays [R,3] x2 and ray_indices / t_starts / t_ends [N] expected")
        pos = torch.empty((n, 3), dtype=torch.float32, device=rays_o.device)
        dirs = torch.empty((n, 3), dtype=torch.float32, device=rays_o.device) if with_dirs else None
        with _Guard(rays_o):
            _check(load_library().nfa_sample_positions(_ptr(rays_o), _ptr(rays_d), rays_o.shape[0], _ptr(ray_indices),
                                                       _ptr(t_starts), _ptr(t_ends), n, _ptr(pos), _ptr(dirs), _stream(rays_o)))
        return (pos, dirs) if with_dirs else pos

    @staticmethod
    def visibility_compact(ray_indices, t_starts, t_ends, dens, from_alpha: bool, early_stop_eps: float,
                           alpha_thre: float, want_mask: bool = False):
        """returns (ray_indices', t_starts', t_ends', mask or None); ONE host sync (the count)."""
        _check_input(ray_indices, "ray_indices", torch.int64)
        for t, nm in ((t_starts, "t_starts"), (t_ends, "t_ends"), (dens, "sigmas/alphas")):
            _check_input(t, nm, torch.float32)
        L = load_library()
        n = dens.shape[0]
        dev = dens.device
        o_idx = torch.empty(n, dtype=torch.int64, device=dev)
        o_t = _rows(2, n, dev)
        mask = torch.empty(n, dtype=torch.bool, device=dev) if want_mask else None
        n_out = _host_ints(dev)
        ws = torch.empty(max(L.nfa_visibility_workspace_bytes(n), 16), dtype=torch.uint8, device=dev)
        with _Guard(dens):
            stream = _stream(dens)
            args = (_ptr(ray_indices), _ptr(t_starts), _ptr(t_ends), _ptr(dens), int(from_alpha), n, early_stop_eps, alpha_thre, _ptr(o_idx),
                    o_t[0].data_ptr(), o_t[1].data_ptr(), _ptr(mask), _ptr(n_out), 0, _ptr(ws))
            # one launch where the call is small enough (mask pass + look-back + compaction, include/nerfacc_hip.h)
            _check(_call("visibility", L.nfa_visibility_compact_sync, *args, _ptr(_sync_block(dev, stream)) if n > 0 else None, stream))
            k = _read_ints(n_out, dev)[0]
            if k < 0:                                    # its look-back gave up (bounded wait): the compaction kernel alone
                _check(L.nfa_visibility_compact_resume(*args, stream))
                k = _read_ints(n_out, dev)[0]
        return o_idx[:k], o_t[0, :k], o_t[1, :k], mask

    @staticmethod
    def accumulate_along_rays(ray_indices, weights, values, n_rays: int, outputs=None):
        _check_input(ray_indices, "ray_indices", torch.int64)
        _check_input(weights, "weights", torch.float32)
        D = 1
        if values is not None:
            _check_input(values, "values", torch.float32)
            D = values.shape[-1]
        if outputs is None:
            outputs = torch.zeros((n_rays, D), dtype=torch.float32, device=weights.device)
        else:
            _check_input(outputs, "outputs", torch.float32)
        with _Guard(weights):
            _check(load_library().nfa_accumulate_along_rays(_ptr(ray_indices), _ptr(weights), _ptr(values), weights.shape[0], D,
                                                            outputs.shape[0], _ptr(outputs), _stream(weights)))
        return outputs

    @staticmethod
    def accumulate_along_rays_bwd(ray_indices, weights, values, g_out, need_w: bool, need_v: bool):
        _check_input(g_out, "g_outputs", torch.float32)
        D = g_out.shape[-1]
        g_w = torch.empty_like(weights) if need_w else None
        g_v = torch.empty_like(values) if (need_v and values is not None) else None
        with _Guard(weights):
            _check(load_library().nfa_accumulate_along_rays_bwd(_ptr(ray_indices), _ptr(weights), _ptr(values), _ptr(g_out),
                                                                weights.shape[0], D, g_out.shape[0], _ptr(g_w), _ptr(g_v), _stream(weights)))
        return g_w, g_v

    # ---- occupancy-grid maintenance (OccGridEstimator._update, occ_grid.py:366-404) ----
    @staticmethod
    def grid_cell_points(cell_ids, jitter, resolution, aabb):
        """world positions of jittered voxels: aabb.lo + ((coords(cell_ids) + jitter) / resolution) * extent.
        cell_ids: int64 [n] level-local ids or None (cells 0..n-1); jitter: float32 [n, 3];
        resolution: 3 host ints; aabb: float32 [6] device tensor of the level."""
        _check_input(jitter, "jitter", torch.float32)
        _check_input(aabb, "aabb", torch.float32)
        n = jitter.shape[0]
        if cell_ids is not None:
            _check_input(cell_ids, "cell_ids", torch.int64)
            if cell_ids.shape != (n,):
                raise RuntimeError("cell_ids must have shape [n] matching jitter [n, 3]")
        if jitter.shape != (n, 3) or aabb.numel() != 6:
            raise RuntimeError("jitter must be [n, 3] and aabb must hold 6 floats")
        rx, ry, rz = (int(r) for r in resolution)
        points = torch.empty((n, 3), dtype=torch.float32, device=jitter.device)
        with _Guard(jitter):
            _check(_call("grid_cell_points", load_library().nfa_grid_cell_points,
                         _ptr(cell_ids), n, _ptr(jitter), rx, ry, rz, _ptr(aabb), _ptr(points), _stream(jitter)))
        return points

    @staticmethod
    def grid_ema_update(occs_level, cell_ids, occ_new, ema_decay: float):
        """in place: occs_level[cell_ids] = maximum(occs_level[cell_ids] * ema_decay, occ_new)"""
        _check_input(occs_level, "occs", torch.float32)
        _check_input(occ_new, "occ_new", torch.float32)
        n = occ_new.numel()
        if cell_ids is not None:
            _check_input(cell_ids, "cell_ids", torch.int64)
            if cell_ids.numel() != n:
                raise RuntimeError("cell_ids and occ_new must have the same number of elements")
        elif n > occs_level.numel():
            raise RuntimeError("occ_new has more elements than the level has cells")
        scratch = torch.empty(n, dtype=torch.float32, device=occs_level.device)
        with _Guard(occs_level):
            _check(_call("grid_ema_update", load_library().nfa_grid_ema_update,
                         _ptr(occs_level), _ptr(cell_ids), n, _ptr(occ_new), float(ema_decay), _ptr(scratch),
                         _stream(occs_level)))

    @staticmethod
    def grid_mark_invisible(occs_level, cell_ids, resolution, aabb, w2c_R, w2c_T, K, width: float, height: float, near_plane: float):
        """in place on one level: occs_level[cell_ids] = 0 where a camera covers the cell (and none is nearer than
        near_plane in front of it), -1 elsewhere (occ_grid.py:262-332).  w2c_R [C,3,3], w2c_T [C,3,1], K [C or 1,3,3]."""
        for t, nm in ((occs_level, "occs"), (aabb, "aabb"), (w2c_R, "w2c_R"), (w2c_T, "w2c_T"), (K, "K")):
            _check_input(t, nm, torch.float32)
        n = occs_level.numel()
        if cell_ids is not None:
            _check_input(cell_ids, "cell_ids", torch.int64)
            n = cell_ids.numel()
        C = w2c_R.shape[0]
        if w2c_R.numel() != 9 * C or w2c_T.numel() != 3 * C or K.numel() not in (9, 9 * C):
            raise RuntimeError("grid_mark_invisible: w2c_R [C,3,3], w2c_T [C,3,1], K [C or 1,3,3] expected")
        rx, ry, rz = (int(r) for r in resolution)
        with _Guard(occs_level):
            _check(load_library().nfa_grid_mark_invisible(_ptr(occs_level), _ptr(cell_ids), n, rx, ry, rz, _ptr(aabb), _ptr(w2c_R),
                                                          _ptr(w2c_T), _ptr(K), C, int(K.numel() == 9 and C != 1), float(width),
                                                          float(height), float(near_plane), _stream(occs_level)))

    @staticmethod
    def grid_occupied_counts(binaries):
        """occupied voxels per level (what nonzero(binaries[level]) would count), from the packed grid's header"""
        _nonempty_bricks(binaries)
        return list(_brick_entry(binaries)["level_counts"])

    @staticmethod
    def grid_occupied_cells(binaries, lvl: int):
        """the occupied cells of level `lvl`, ascending — torch.nonzero(binaries[lvl].flatten())[:, 0] (occ_grid.py:356) — one launch,
        no read-back (the count is in the packed grid's header)"""
        _check_input(binaries, "binaries", torch.bool)
        cnt = _CtypesC.grid_occupied_counts(binaries)[lvl]
        out = torch.empty(cnt, dtype=torch.int64, device=binaries.device)
        if cnt:
            n_cells = binaries[lvl].numel()
            with _Guard(binaries):
                stream = _stream(binaries)
                _check(load_library().nfa_grid_occupied_cells(binaries.data_ptr() + lvl * n_cells, n_cells, _ptr(out), cnt,
                                                              _ptr(_sync_block(binaries.device, stream)), stream))
        return out

    @staticmethod
    def grid_threshold(occs, occ_thre: float, shape=None):
        """binaries (bool) = occs > min(mean(occs[occs >= 0]), occ_thre); also returns the threshold as a 1-element
        device tensor (no host sync).  shape = None: flat grid.  shape = (G, rx, ry, rz): the grid in that shape and,
        from the same pass, its bit-packed form, entered into the brick cache (the next traversal does not pack again)."""
        _check_input(occs, "occs", torch.float32)
        L = load_library()
        n = occs.numel()
        ws = torch.empty(L.nfa_grid_threshold_workspace_bytes() // 8, dtype=torch.float64, device=occs.device)
        thre = torch.empty(1, dtype=torch.float32, device=occs.device)
        if shape is not None:
            G, rx, ry, rz = (int(x) for x in shape)
            if G * rx * ry * rz != n or n == 0:
                raise RuntimeError("grid_threshold: shape must be (n_grids, resx, resy, resz) with as many cells as occs")
            binaries = torch.empty((G, rx, ry, rz), dtype=torch.bool, device=occs.device)
            bricks = torch.empty(L.nfa_packed_grid_words(G, rx, ry, rz), dtype=torch.int64, device=occs.device)
            with _Guard(occs):
                _check(_call("grid_threshold", L.nfa_grid_threshold_packed, _ptr(occs), G, rx, ry, rz, float(occ_thre), _ptr(ws),
                             _ptr(binaries), _ptr(thre), _ptr(bricks), _stream(occs)))
                with _brick_lock:
                    _insert_brick_entry(binaries, bricks)
            return binaries, thre
        binaries = torch.empty(n, dtype=torch.bool, device=occs.device)
        with _Guard(occs):
            _check(_call("grid_threshold", L.nfa_grid_threshold, _ptr(occs), n, float(occ_thre), _ptr(ws),
                         _ptr(binaries), _ptr(thre), _stream(occs)))
        return binaries, thre

    @staticmethod
    def rendering_fwd(ray_indices, t_starts, t_ends, sigmas, rgbs, n_rays: int, bkgd, expected_depths: bool):
        _check_input(ray_indices, "ray_indices", torch.int64)
        for t, nm in ((t_starts, "t_starts"), (t_ends, "t_ends"), (sigmas, "sigmas"), (rgbs, "rgbs")):
            _check_input(t, nm, torch.float32)
        if bkgd is not None:
            _check_input(bkgd, "render_bkgd", torch.float32)
        n = sigmas.shape[0]
        dev = sigmas.device
        per_sample = _rows(3, n, dev)
        colors = torch.empty((n_rays, 3), dtype=torch.float32, device=dev)
        od = torch.empty((2, n_rays, 1), dtype=torch.float32, device=dev)
        with _Guard(sigmas):
            _check(_call("rendering_fwd", load_library().nfa_rendering_fwd,
                _ptr(ray_indices), _ptr(t_starts), _ptr(t_ends), _ptr(sigmas), _ptr(rgbs), n, n_rays, _ptr(bkgd),
                int(expected_depths), per_sample[0].data_ptr(), per_sample[1].data_ptr(), per_sample[2].data_ptr(),
                _ptr(colors), od[0].data_ptr(), od[1].data_ptr(), _stream(sigmas)))
        return colors, od[0], od[1], per_sample[0], per_sample[1], per_sample[2]

    @staticmethod
    def rendering_bwd(ray_indices, t_starts, t_ends, sigmas, rgbs, weights, trans, alphas, opacities, depths,
                      n_rays: int, bkgd, expected_depths: bool, g_colors, g_opac, g_depth, g_w, g_T, g_a,
                      need_sigma: bool = True, need_rgb: bool = True):
        for t, nm in ((g_colors, "g_colors"), (g_opac, "g_opacities"), (g_depth, "g_depths"), (g_w, "g_weights"),
                      (g_T, "g_trans"), (g_a, "g_alphas")):
            if t is not None:
                _check_input(t, nm, torch.float32)
        g_sig = torch.empty_like(sigmas) if need_sigma else None
        g_rgb = torch.empty_like(rgbs) if need_rgb else None
        with _Guard(sigmas):
            _check(_call("rendering_bwd", load_library().nfa_rendering_bwd,
                _ptr(ray_indices), _ptr(t_starts), _ptr(t_ends), _ptr(sigmas), _ptr(rgbs), _ptr(weights), _ptr(trans),
                _ptr(alphas), _ptr(opacities), _ptr(depths), sigmas.shape[0], n_rays, _ptr(bkgd), int(expected_depths),
                _ptr(g_colors), _ptr(g_opac), _ptr(g_depth), _ptr(g_w), _ptr(g_T), _ptr(g_a), _ptr(g_sig), _ptr(g_rgb),
                _stream(sigmas)))
        return g_sig, g_rgb


def _select_backend():
    """The torch C++ extension (nerfacc_amd/_hip*.so, csrc/torch_ext.cpp) is THE boundary, as the reference's pybind
    module is; the ctypes face of the same C ABI stays as a fallback (`NERFACC_AMD_BACKEND=ctypes` forces it)."""
    want = os.environ.get("NERFACC_AMD_BACKEND", "ext").lower()
    if want not in ("ext", "ctypes"):
        raise ImportError(f"NERFACC_AMD_BACKEND must be 'ext' or 'ctypes', got {want!r}")
    if want == "ext" and os.environ.get("NERFACC_AMD_LIB"):
        # an explicitly named library (instrumented / variant builds) is only honoured by the ctypes face: the extension
        # links the in-tree library through its rpath
        want = "ctypes"
    if want == "ext":
        import glob
        import importlib
        import warnings

        if not glob.glob(os.path.join(_PKG, "_hip*.so")):
            _jit_build()
        if glob.glob(os.path.join(_PKG, "_hip*.so")):
            try:
                return importlib.import_module("nerfacc_amd._hip"), "ext"
            except ImportError as e:           # a stale build (another torch / ABI): say so and use the other face
                warnings.warn(f"nerfacc_amd: the torch extension nerfacc_amd/_hip*.so does not import ({e}); using the ctypes "
                              "backend (python -m nerfacc_amd.build --force rebuilds it)")
        else:
            warnings.warn("nerfacc_amd: the torch extension nerfacc_amd/_hip*.so is not built; using the ctypes backend "
                          "(python -m nerfacc_amd.build builds both)")
    load_library()
    return _CtypesC, "ctypes"


BACKEND = "ctypes"          # set for real just below (the functions above read it at call time)
_C, BACKEND = _select_backend()
if BACKEND == "ext":
    # the extension keeps device tensors in thread-local slots (traversal workspace, a count pass launched ahead): hand the main
    # thread's back while the interpreter and the HIP runtime are still up (destruction order at shutdown)
    import atexit

    atexit.register(lambda: getattr(_C, "release_workspace", lambda: None)())
RaySegmentsSpec = _C.RaySegmentsSpec

__all__ = ["_C", "BACKEND", "KernelTimer", "set_kernel_timer", "load_library", "LIB_PATH", "EXPORTED_SYMBOLS", "RaySegmentsSpec", "packed_bricks"]
