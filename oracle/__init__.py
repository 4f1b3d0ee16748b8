"""CPU oracle for the nerfacc OccGrid sampling + rendering hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; nothing under
``nerfacc_amd/`` does.  The arithmetic lives in ``nerfacc_oracle.c`` (plain C,
single-threaded, one reference citation per function); this module is the
numpy/ctypes face of it plus the small amount of host logic the reference keeps
in C++/Python around its kernels (allocation from counts, two-pass traversal,
mask compaction).

Parity status: pinned against the golden fixtures in ``tests/golden`` (made by
``tests/golden/make_golden.py`` from the importable, pure-torch parts of the
reference), the reference tests' hand-computed answers, and — for the exact
sample lists of ``traverse_grids`` — ``tests/golden/k2_reference.npz``, produced
by the reference's own ``grid.cu`` compiled for the host (``oracle/ref_shim`` ->
``oracle/_ref``, see ``tests/golden/make_k2_golden.py`` and DESIGN.md §3.4/§4).

``oracle.torch_cpu`` holds the pure-PyTorch CPU composition of the rendering ops
(the reference's batched branch), timed by bench.py's ``cpu_baseline`` leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile liborc.so with gcc (seconds).  Returns the library path."""
    src = os.path.join(_HERE, "nerfacc_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liborc.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_threads(n: int) -> int:
    """OpenMP threads used by the traversal / rendering entry points (default 1: the checker is scalar and
    single-threaded; bench.py's all-cores CPU baseline raises it).  Results do not depend on it."""
    lib().orc_set_threads(int(n))
    return int(lib().orc_get_threads())


# ---------------------------------------------------------------- helpers
def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _i64(x):
    return np.ascontiguousarray(x, dtype=np.int64)


def _u8(x):
    return np.ascontiguousarray(np.asarray(x).astype(np.uint8))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


_c_f = ctypes.c_float
_c_i32 = ctypes.c_int32
_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int


# ---------------------------------------------------------------- K1
def ray_aabb_intersect(rays_o, rays_d, aabbs, near=-np.inf, far=np.inf, miss=np.inf):
    """nerfacc/grid.py:13-51 -> grid.cu:477-519."""
    rays_o, rays_d, aabbs = _f32(rays_o), _f32(rays_d), _f32(aabbs)
    R, G = rays_o.shape[0], aabbs.shape[0]
    t_mins = np.empty((R, G), np.float32)
    t_maxs = np.empty((R, G), np.float32)
    hits = np.empty((R, G), np.uint8)
    lib().orc_ray_aabb_intersect(
        _c_i64(R), _p(rays_o), _p(rays_d), _c_i64(G), _p(aabbs),
        _c_f(near), _c_f(far), _c_f(miss), _p(t_mins), _p(t_maxs), _p(hits))
    return t_mins, t_maxs, hits.astype(bool)


# ---------------------------------------------------------------- K2
def traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes=None, far_planes=None,
                   step_size=1e-3, cone_angle=0.0, traverse_steps_limit=None,
                   over_allocate=False, rays_mask=None, t_sorted=None, t_indices=None, hits=None):
    """nerfacc/grid.py:93-192 + grid.cu:320-474 (host logic) around K2.

    Returns (intervals, samples, terminate_planes) where intervals/samples are
    dicts with the fields of RaySegmentsSpec (data_spec.hpp:6-14).
    """
    rays_o, rays_d, aabbs = _f32(rays_o), _f32(rays_d), _f32(aabbs)
    binaries = _u8(binaries)
    R = rays_o.shape[0]
    G = binaries.shape[0]
    res = np.asarray(binaries.shape[1:], np.int32)
    near = _f32(np.zeros(R) if near_planes is None else near_planes)
    far = _f32(np.full(R, np.inf) if far_planes is None else far_planes)
    mask = _u8(np.ones(R, bool) if rays_mask is None else rays_mask)
    limit = -1 if traverse_steps_limit is None else int(traverse_steps_limit)
    if over_allocate:
        assert limit > 0
    if t_sorted is None or t_indices is None or hits is None:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        cat = np.concatenate([t_mins, t_maxs], -1)
        # torch.sort is not guaranteed stable; ties only matter for degenerate
        # rays and a stable sort is what torch's CUDA sort yields for 2G <= 32.
        t_indices = np.argsort(cat, axis=-1, kind="stable").astype(np.int64)
        t_sorted = np.take_along_axis(cat, t_indices, -1)
    t_sorted, t_indices, hits = _f32(t_sorted), _i64(t_indices), _u8(hits)

    common = (_c_i64(R), _p(rays_o), _p(rays_d))
    grid = (_c_i32(G), _p(res), _p(binaries), _p(aabbs), _p(hits), _p(t_sorted), _p(t_indices),
            _p(near), _p(far), _c_f(step_size), _c_f(cone_angle), _c_i32(limit))
    term = np.empty(R, np.float32)
    L = lib()
    if over_allocate:
        iv_cnts = (np.full(R, limit * 2, np.int64) * mask)
        sm_cnts = (np.full(R, limit, np.int64) * mask)
        use_mask = _p(mask)
    else:
        iv_cnts = np.empty(R, np.int64)
        sm_cnts = np.empty(R, np.int64)
        L.orc_traverse_count(*common, None, *grid, _p(iv_cnts), _p(sm_cnts), None)
        use_mask = None
    iv_starts = np.cumsum(iv_cnts) - iv_cnts
    sm_starts = np.cumsum(sm_cnts) - sm_cnts
    E, N = int(iv_cnts.sum()), int(sm_cnts.sum())
    iv = dict(vals=np.zeros(E, np.float32), ray_indices=np.zeros(E, np.int64),
              is_left=np.zeros(E, np.uint8), is_right=np.zeros(E, np.uint8))
    sm = dict(vals=np.zeros(N, np.float32), ray_indices=np.zeros(N, np.int64),
              is_valid=np.zeros(N, np.uint8))
    if over_allocate:
        # grid.cu:100 — masked-out rays write nothing, not even terminate_planes
        term[:] = 0
    L.orc_traverse_fill(*common, use_mask, *grid,
                        _p(iv_starts), _p(iv_cnts), _p(sm_starts), _p(sm_cnts),
                        _p(iv["vals"]), _p(iv["ray_indices"]), _p(iv["is_left"]), _p(iv["is_right"]),
                        _p(sm["vals"]), _p(sm["ray_indices"]), _p(sm["is_valid"]), _p(term), None, None)
    if over_allocate:
        # grid.cu:402-404: chunk_starts recomputed from the ACTUAL counts
        iv_starts = np.cumsum(iv_cnts) - iv_cnts
        sm_starts = np.cumsum(sm_cnts) - sm_cnts
    for d, s, c in ((iv, iv_starts, iv_cnts), (sm, sm_starts, sm_cnts)):
        d["chunk_starts"], d["chunk_cnts"] = s, c
        d["packed_info"] = np.stack([s, c], -1)
    for k in ("is_left", "is_right"):
        iv[k] = iv[k].astype(bool)
    sm["is_valid"] = sm["is_valid"].astype(bool)
    return iv, sm, term


def sampling(rays_o, rays_d, binaries, aabbs, near_plane=0.0, far_plane=1e10, t_min=None,
             t_max=None, render_step_size=1e-3, cone_angle=0.0, sigmas_fn=None,
             early_stop_eps=1e-4, alpha_thre=0.0, occs_mean=None, jitter=None):
    """OccGridEstimator.sampling, nerfacc/estimators/occ_grid.py:154-221.

    ``jitter`` stands in for torch.rand_like(near_planes) (stratified); the
    caller supplies it so both sides see the same numbers.
    """
    R = np.asarray(rays_o).shape[0]
    near = np.full(R, near_plane, np.float32)
    far = np.full(R, far_plane, np.float32)
    if t_min is not None:
        near = np.maximum(near, _f32(t_min))
    if t_max is not None:
        far = np.minimum(far, _f32(t_max))
    if jitter is not None:
        near = (near + _f32(jitter) * np.float32(render_step_size)).astype(np.float32)
    iv, sm, _ = traverse_grids(rays_o, rays_d, binaries, aabbs, near, far, render_step_size, cone_angle)
    t_starts = iv["vals"][iv["is_left"]]
    t_ends = iv["vals"][iv["is_right"]]
    ray_indices = sm["ray_indices"]
    if (alpha_thre > 0.0 or early_stop_eps > 0.0) and sigmas_fn is not None:
        if occs_mean is not None:
            alpha_thre = min(alpha_thre, occs_mean)
        sigmas = _f32(sigmas_fn(t_starts, t_ends, ray_indices))
        _, trans, alphas = render_weight_from_density(t_starts, t_ends, sigmas, ray_indices)
        keep = visibility(trans, alphas, early_stop_eps, alpha_thre)
        ray_indices, t_starts, t_ends = ray_indices[keep], t_starts[keep], t_ends[keep]
    return ray_indices, t_starts, t_ends, sm["packed_info"]


def sample_occgrid(rays_o, rays_d, binaries, aabbs, near, far, step_size, cone_angle=0.0):
    """traverse_grids + the two boolean-mask gathers of occ_grid.py:174-176, with each sample's interval written by
    the fill pass itself (threaded when set_threads > 1; no edge arrays, no serial numpy passes):
    (ray_indices, t_starts, t_ends, packed_info).  Equal to `traverse_grids` followed by `vals[is_left]`,
    `vals[is_right]` (tests/test_oracle.py)."""
    rays_o, rays_d, aabbs = _f32(rays_o), _f32(rays_d), _f32(aabbs)
    binaries = _u8(binaries)
    R, G = rays_o.shape[0], binaries.shape[0]
    res = np.asarray(binaries.shape[1:], np.int32)
    near, far = _f32(near), _f32(far)
    t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
    cat = np.concatenate([t_mins, t_maxs], -1)
    t_indices = np.argsort(cat, axis=-1, kind="stable").astype(np.int64)
    t_sorted = _f32(np.take_along_axis(cat, t_indices, -1))
    hits = _u8(hits)
    common = (_c_i64(R), _p(rays_o), _p(rays_d))
    grid = (_c_i32(G), _p(res), _p(binaries), _p(aabbs), _p(hits), _p(t_sorted), _p(t_indices),
            _p(near), _p(far), _c_f(step_size), _c_f(cone_angle), _c_i32(-1))
    sm_cnts = np.empty(R, np.int64)
    L = lib()
    L.orc_traverse_count(*common, None, *grid, None, _p(sm_cnts), None)
    sm_starts = np.cumsum(sm_cnts) - sm_cnts
    n = int(sm_cnts.sum())
    ri, ts, te = np.empty(n, np.int64), np.empty(n, np.float32), np.empty(n, np.float32)
    L.orc_traverse_fill(*common, None, *grid, None, None, _p(sm_starts), _p(sm_cnts),
                        None, None, None, None, None, _p(ri), None, None, _p(ts), _p(te))
    return ri, ts, te, np.stack([sm_starts, sm_cnts], -1)


def compact(keep, ray_indices, t_starts, t_ends, packed_info):
    """x = x[keep] for the three sample arrays (occ_grid.py:218-220), per ray in C; returns the new packed_info too"""
    keep = _u8(keep)
    starts, cnts = _i64(packed_info[:, 0]), _i64(packed_info[:, 1])
    R = starts.shape[0]
    kept = np.empty(R, np.int64)
    lib().orc_count_kept(_c_i64(R), _p(starts), _p(cnts), _p(keep), _p(kept))
    out_starts = np.cumsum(kept) - kept
    n = int(kept.sum())
    ri, ts, te = np.empty(n, np.int64), np.empty(n, np.float32), np.empty(n, np.float32)
    lib().orc_compact_samples(_c_i64(R), _p(starts), _p(cnts), _p(keep), _p(out_starts), _p(_i64(ray_indices)),
                              _p(_f32(t_starts)), _p(_f32(t_ends)), _p(ri), _p(ts), _p(te))
    return ri, ts, te, np.stack([out_starts, kept], -1)


# ---------------------------------------------------------------- pack / scans
def pack_info(ray_indices, n_rays):
    ray_indices = _i64(ray_indices)
    out = np.empty((n_rays, 2), np.int64)
    lib().orc_pack_info(_c_i64(ray_indices.shape[0]), _p(ray_indices), _c_i64(n_rays), _p(out))
    return out


def scan_packed(inputs, packed_info, op="sum", inclusive=True, reverse=False, normalize=False):
    """scan.py packed_info mode (utils_scan.cuh); reverse=True is the backward pass of sums."""
    inputs = _f32(inputs)
    packed_info = _i64(packed_info)
    starts, cnts = _i64(packed_info[:, 0]), _i64(packed_info[:, 1])
    out = np.zeros_like(inputs)
    lib().orc_scan_packed(_c_i64(starts.shape[0]), _p(starts), _p(cnts), _p(inputs), _p(out),
                          _c_int(op == "prod"), _c_int(inclusive), _c_int(reverse), _c_int(normalize))
    return out


def scan_keyed(inputs, indices, op="sum", inclusive=True, reverse=False):
    """scan.py indices mode (scan_cub.cu)."""
    inputs, indices = _f32(inputs), _i64(indices)
    out = np.empty_like(inputs)
    lib().orc_scan_keyed(_c_i64(inputs.shape[0]), _p(indices), _p(inputs), _p(out),
                         _c_int(op == "prod"), _c_int(inclusive), _c_int(reverse))
    return out


def prod_backward(inputs, outputs, grad_outputs, indices, inclusive):
    """scan.cu:199-210 / scan_cub.cu:184-218: revscan(g*out) / clamp_min(in, 1e-10)."""
    g = _f32(grad_outputs) * _f32(outputs)
    rs = scan_keyed(g, indices, "sum", inclusive=inclusive, reverse=True)
    return rs / np.maximum(_f32(inputs), np.float32(1e-10))


# ---------------------------------------------------------------- volrend
def render_weight_from_density(t_starts, t_ends, sigmas, ray_indices, prefix_trans=None):
    t_starts, t_ends, sigmas, ray_indices = _f32(t_starts), _f32(t_ends), _f32(sigmas), _i64(ray_indices)
    n = sigmas.shape[0]
    w, T, a = (np.empty(n, np.float32) for _ in range(3))
    pt = None if prefix_trans is None else _f32(prefix_trans)
    lib().orc_render_weight_from_density(_c_i64(n), _p(ray_indices), _p(t_starts), _p(t_ends),
                                         _p(sigmas), _p(pt), _p(w), _p(T), _p(a))
    return w, T, a


def render_weight_from_density_bwd(t_starts, t_ends, sigmas, ray_indices, g_w=None, g_T=None,
                                   g_a=None, prefix_trans=None):
    t_starts, t_ends, sigmas, ray_indices = _f32(t_starts), _f32(t_ends), _f32(sigmas), _i64(ray_indices)
    n = sigmas.shape[0]
    gs = np.empty(n, np.float32)
    gw = None if g_w is None else _f32(g_w)
    gT = None if g_T is None else _f32(g_T)
    ga = None if g_a is None else _f32(g_a)
    pt = None if prefix_trans is None else _f32(prefix_trans)
    lib().orc_render_weight_from_density_bwd(_c_i64(n), _p(ray_indices), _p(t_starts), _p(t_ends),
                                             _p(sigmas), _p(pt), _p(gw), _p(gT), _p(ga), _p(gs))
    return gs


def render_weight_from_alpha(alphas, ray_indices, prefix_trans=None):
    alphas, ray_indices = _f32(alphas), _i64(ray_indices)
    n = alphas.shape[0]
    w, T = np.empty(n, np.float32), np.empty(n, np.float32)
    pt = None if prefix_trans is None else _f32(prefix_trans)
    lib().orc_render_weight_from_alpha(_c_i64(n), _p(ray_indices), _p(alphas), _p(pt), _p(w), _p(T))
    return w, T


def visibility(trans, alphas, early_stop_eps=1e-4, alpha_thre=0.0):
    trans, alphas = _f32(trans), _f32(alphas)
    out = np.empty(trans.shape[0], np.uint8)
    lib().orc_visibility(_c_i64(trans.shape[0]), _p(trans), _p(alphas), _c_f(early_stop_eps),
                         _c_f(alpha_thre), _p(out))
    return out.astype(bool)


def accumulate_along_rays(weights, values, ray_indices, n_rays, out=None):
    weights, ray_indices = _f32(weights), _i64(ray_indices)
    D = 1 if values is None else int(np.asarray(values).shape[-1])
    vals = None if values is None else _f32(values)
    if out is None:
        out = np.zeros((n_rays, D), np.float32)
    lib().orc_accumulate_along_rays(_c_i64(weights.shape[0]), _p(ray_indices), _p(weights), _p(vals),
                                    _c_i64(D), _p(out))
    return out


def rendering(t_starts, t_ends, ray_indices, n_rays, sigmas, rgbs, render_bkgd=None,
              expected_depths=True):
    t_starts, t_ends, sigmas, rgbs = _f32(t_starts), _f32(t_ends), _f32(sigmas), _f32(rgbs)
    ray_indices = _i64(ray_indices)
    n = sigmas.shape[0]
    w, T, a = (np.empty(n, np.float32) for _ in range(3))
    colors = np.empty((n_rays, 3), np.float32)
    opac = np.empty((n_rays, 1), np.float32)
    depth = np.empty((n_rays, 1), np.float32)
    bk = None if render_bkgd is None else _f32(render_bkgd)
    lib().orc_rendering(_c_i64(n), _c_i64(n_rays), _p(ray_indices), _p(t_starts), _p(t_ends),
                        _p(sigmas), _p(rgbs), _p(bk), _c_int(expected_depths),
                        _p(w), _p(T), _p(a), _p(colors), _p(opac), _p(depth))
    return colors, opac, depth, dict(weights=w, trans=T, alphas=a)


# ---------------------------------------------------------------- pdf
def importance_sampling(vals, cdfs, n_intervals_per_ray):
    vals, cdfs = _f32(vals), _f32(cdfs)
    R, E = vals.shape
    n = int(n_intervals_per_ray)
    edges = np.zeros((R, n + 1), np.float32)
    mids = np.zeros((R, n), np.float32)
    lib().orc_importance_sampling_batched(_c_i64(R), _c_i64(E), _p(vals), _p(cdfs), _c_i64(n),
                                          _p(edges), _p(mids))
    return edges, mids


def searchsorted(key_vals, query_vals):
    key_vals, query_vals = _f32(key_vals), _f32(query_vals)
    R, K = key_vals.shape
    Q = query_vals.shape[1]
    l = np.empty((R, Q), np.int64)
    r = np.empty((R, Q), np.int64)
    lib().orc_searchsorted_batched(_c_i64(R), _c_i64(Q), _p(query_vals), _c_i64(K), _p(key_vals),
                                   _p(l), _p(r))
    return l, r


# ---------------------------------------------------------------- occupancy-grid maintenance
# numpy restatement of OccGridEstimator._update (nerfacc/estimators/occ_grid.py:366-404).  The
# random numbers are inputs (the reference draws them with torch's generator); float32
# arithmetic in the reference's operation order.
def grid_cell_points(cell_ids, jitter, resolution, aabb):
    """occ_grid.py:377-384: x = (grid_coords[ids] + rand) / resolution; x = lo + x * (hi - lo)."""
    jitter, aabb = _f32(jitter), _f32(aabb)
    rx, ry, rz = (int(r) for r in resolution)
    ids = np.arange(jitter.shape[0], dtype=np.int64) if cell_ids is None else _i64(cell_ids)
    coords = np.stack([ids // (ry * rz), (ids // rz) % ry, ids % rz], -1).astype(np.float32)   # x-major (grid.cu:187-192)
    unit = (coords + jitter) / np.array([rx, ry, rz], np.float32)
    return (aabb[:3] + unit * (aabb[3:] - aabb[:3])).astype(np.float32)


def grid_ema_update(occs_level, cell_ids, occ_new, ema_decay=0.95):
    """occ_grid.py:388-390: occs[ids] = maximum(occs[ids] * decay, occ); returns the new level.
    Repeated ids: numpy (like CPU index_put_) keeps the last candidate; any candidate is a valid
    outcome of the reference's GPU scatter — compare with `grid_ema_candidates` there."""
    occs_level = _f32(occs_level).copy()
    ids = np.arange(len(occ_new), dtype=np.int64) if cell_ids is None else _i64(cell_ids)
    occs_level[ids] = np.maximum(occs_level[ids] * np.float32(ema_decay), _f32(occ_new))
    return occs_level


def grid_ema_candidates(occs_level, cell_ids, occ_new, ema_decay=0.95):
    """the value each (id, occ) pair wants to write, formed from the OLD grid"""
    ids = np.arange(len(occ_new), dtype=np.int64) if cell_ids is None else _i64(cell_ids)
    return np.maximum(_f32(occs_level)[ids] * np.float32(ema_decay), _f32(occ_new))


def grid_threshold(occs, occ_thre=0.01):
    """occ_grid.py:392-404: thre = clamp(mean(occs[occs >= 0]), max=occ_thre); binaries = occs > thre.
    The mean is accumulated in float64 (torch's float32 tree sum differs from any fixed order in
    the last bits; cells that close to the threshold are excluded from parity checks)."""
    occs = _f32(occs)
    vis = occs[occs >= 0]
    mean = np.float32(vis.astype(np.float64).mean()) if vis.size else np.float32(np.nan)
    thre = mean if np.isnan(mean) else np.float32(min(mean, np.float32(occ_thre)))
    return occs > thre, thre
