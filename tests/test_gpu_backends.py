"""The two host faces of the C ABI — the torch C++ extension (default; nerfacc_amd/_hip*.so) and the ctypes fallback —
must return identical tensors for identical inputs: they call the same kernels with the same arguments."""
import numpy as np
import pytest
import torch

from gpu_utils import DEV, lego_like, ragged, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def faces():
    from nerfacc_amd.cuda import _backend

    if _backend.BACKEND != "ext":
        pytest.skip("torch extension not the active backend")
    return _backend._C, _backend._CtypesC


def _same(a, b):
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y)
    elif a is None or b is None:
        assert a is None and b is None
    elif torch.is_tensor(a):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
    else:
        for k in ("vals", "is_left", "is_right", "is_valid", "chunk_starts", "chunk_cnts", "ray_indices"):
            _same(getattr(a, k), getattr(b, k))


def test_sampling_traversal_and_rendering_agree(faces):
    ext, ct = faces
    o, d, aabb, occ = lego_like(5, 3000, res=64)
    O, D, B, A = t(o), t(d), t(occ), t(aabb)
    R = O.shape[0]
    near, far = torch.zeros(R, device=DEV), torch.full((R,), 1e10, device=DEV)
    a = ext.sample_occgrid(O, D, B, A, near, far, 5e-3, 0.0)
    b = ct.sample_occgrid(O, D, B, A, near, far, 5e-3, 0.0)
    _same(a, b)
    mask = torch.rand(R, device=DEV) < 0.5
    _same(ext.sample_occgrid(O, D, B, A, near, far, 5e-3, 0.0, rays_mask=mask, traverse_steps_limit=6, with_terminate_planes=True),
          ct.sample_occgrid(O, D, B, A, near, far, 5e-3, 0.0, rays_mask=mask, traverse_steps_limit=6, with_terminate_planes=True))
    ones = torch.ones(R, dtype=torch.bool, device=DEV)
    for over, limit in ((False, -1), (True, 5)):
        _same(ext.traverse_grids(O, D, ones, B, A, None, None, None, near, far, 5e-3, 0.0, True, True, True, limit, over),
              ct.traverse_grids(O, D, ones, B, A, None, None, None, near, far, 5e-3, 0.0, True, True, True, limit, over))
    _same(ext.ray_aabb_intersect(O, D, A, 0.0, 10.0, -1.0), ct.ray_aabb_intersect(O, D, A, 0.0, 10.0, -1.0))
    ri, ts, te, pk = a
    sig = torch.rand(ri.shape[0], device=DEV) * 8
    rgb = torch.rand(ri.shape[0], 3, device=DEV)
    _same(ext.visibility_compact(ri, ts, te, sig, False, 1e-3, 0.01, True), ct.visibility_compact(ri, ts, te, sig, False, 1e-3, 0.01, True))
    bk = torch.ones(3, device=DEV)
    fa, fb = ext.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True), ct.rendering_fwd(ri, ts, te, sig, rgb, R, bk, True)
    _same(fa, fb)
    g = [torch.rand_like(x) for x in fa]
    args = (ri, ts, te, sig, rgb, fa[3], fa[4], fa[5], fa[1], fa[2], R, bk, True, *g)
    _same(ext.rendering_bwd(*args), ct.rendering_bwd(*args))
    _same(ext.render_weight_from_density_fwd(ri, ts, te, sig), ct.render_weight_from_density_fwd(ri, ts, te, sig))
    _same(ext.sample_positions(O, D, ri, ts, te, True), ct.sample_positions(O, D, ri, ts, te, True))
    _same(ext.pack_info(ri, R), ct.pack_info(ri, R))
    _same(ext.accumulate_along_rays(ri, sig, rgb, R), ct.accumulate_along_rays(ri, sig, rgb, R))
    _same(ext.accumulate_along_rays_bwd(ri, sig, rgb, fa[0], True, True), ct.accumulate_along_rays_bwd(ri, sig, rgb, fa[0], True, True))


def test_scans_and_pdf_agree(faces):
    ext, ct = faces
    rng = np.random.default_rng(0)
    ri_np, pk_np = ragged(rng, 500, 40)
    ri, pk = t(ri_np), t(pk_np)
    x = torch.rand(ri.shape[0], device=DEV) + 0.5
    s, c = pk[:, 0].contiguous(), pk[:, 1].contiguous()
    for name in ("inclusive_sum", "exclusive_sum"):
        _same(getattr(ext, name)(s, c, x, False, False), getattr(ct, name)(s, c, x, False, False))
        _same(getattr(ext, name)(s, c, x, True, False), getattr(ct, name)(s, c, x, True, False))     # normalize
        _same(getattr(ext, name)(s, c, x, False, True), getattr(ct, name)(s, c, x, False, True))     # backward
        with pytest.raises(RuntimeError):                                                             # scan.cu:25-26
            getattr(ext, name)(s, c, x, True, True)
        _same(getattr(ext, name + "_cub")(ri, x, False), getattr(ct, name + "_cub")(ri, x, False))
    for name in ("inclusive_prod", "exclusive_prod"):
        y = getattr(ext, name + "_forward")(s, c, x)
        _same(y, getattr(ct, name + "_forward")(s, c, x))
        _same(getattr(ext, name + "_cub_forward")(ri, x), getattr(ct, name + "_cub_forward")(ri, x))
        g = torch.rand_like(x)
        _same(getattr(ext, name + "_backward")(s, c, x, y, g), getattr(ct, name + "_backward")(s, c, x, y, g))
        _same(getattr(ext, name + "_cub_backward")(ri, x, y, g), getattr(ct, name + "_cub_backward")(ri, x, y, g))
    vals = torch.sort(torch.rand(64, 33, device=DEV), -1)[0].contiguous()
    cdfs = torch.sort(torch.rand(64, 33, device=DEV), -1)[0].contiguous()
    cdfs[:, 0], cdfs[:, -1] = 0.0, 1.0
    se, sc = ext.RaySegmentsSpec(), ct.RaySegmentsSpec()
    se.vals = sc.vals = vals
    (ie, me), (ic, mc) = ext.importance_sampling(se, cdfs, 16, False), ct.importance_sampling(sc, cdfs, 16, False)
    _same((ie.vals, me.vals), (ic.vals, mc.vals))
    qe, qc = ext.RaySegmentsSpec(), ct.RaySegmentsSpec()
    qe.vals = qc.vals = ie.vals
    _same(ext.searchsorted(qe, se), ct.searchsorted(qc, sc))


def test_kernel_timer_through_the_extension(faces):
    from nerfacc_amd.cuda import _backend

    ext, _ = faces
    o, d, aabb, occ = lego_like(2, 2000, res=64)
    O, D, B, A = t(o), t(d), t(occ), t(aabb)
    near, far = torch.zeros(2000, device=DEV), torch.full((2000,), 1e10, device=DEV)
    timer = _backend.KernelTimer(names=("traverse_count", "traverse_fill"))
    _backend.set_kernel_timer(timer)
    try:
        for _ in range(3):
            ext.sample_occgrid(O, D, B, A, near, far, 5e-3, 0.0)
    finally:
        summ = timer.summary()
        _backend.set_kernel_timer(None)
    assert summ["traverse_count"][0] == 3 and 0.0 < summ["traverse_count"][1] < 5.0
    # (the emit pass is launched speculatively from the second call on; a guess left over from an earlier test that is too
    # small for these rays adds one relaunch)
    assert 3 <= summ["traverse_fill"][0] <= 4


def test_extension_rejects_mismatched_arguments(faces):
    """lengths, dtypes and devices are checked on the host before any kernel sees a pointer (the reference's CHECK_INPUT
    checks device + contiguity only, utils_cuda.cuh:12-17; a short tensor there is an out-of-bounds read)"""
    ext, _ = faces
    n = 1000
    ri = torch.zeros(n, dtype=torch.int64, device=DEV)
    f = torch.rand(n, device=DEV)
    rgb = torch.rand(n, 3, device=DEV)
    with pytest.raises(RuntimeError, match="t_ends"):
        ext.render_weight_from_density_fwd(ri, f, f[:-1].contiguous(), f, None)
    with pytest.raises(RuntimeError, match="ray_indices"):
        ext.render_weight_from_density_fwd(ri.int(), f, f, f, None)
    with pytest.raises(RuntimeError, match="rgbs"):
        ext.rendering_fwd(ri, f, f, f, rgb[:-1].contiguous(), 1, None, True)
    with pytest.raises(RuntimeError, match="render_bkgd"):
        ext.rendering_fwd(ri, f, f, f, rgb, 1, torch.ones(4, device=DEV), True)
    with pytest.raises(RuntimeError, match="values"):
        ext.accumulate_along_rays(ri, f, rgb[:-1].contiguous(), 1)
    with pytest.raises(RuntimeError, match="outputs"):
        ext.accumulate_along_rays(ri, f, rgb, 1, torch.zeros(1, 2, device=DEV))
    with pytest.raises(RuntimeError, match="CUDA/HIP"):
        ext.render_weight_from_density_fwd(ri.cpu(), f, f, f, None)
    with pytest.raises(RuntimeError, match="t_starts"):
        ext.visibility_compact(ri, f[:-1].contiguous(), f, f, False, 1e-4, 0.0)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.exclusive_sum_cub(ri, torch.rand(n, 2, device=DEV)[:, 0], False)
