"""HIP traversal / ray-AABB kernels vs the CPU oracle: bit-exact (integer AND float outputs,
both sides use the same op order and fmaf sites).  Calls go through the public API, i.e.
through the C ABI of libnerfacc_hip.so."""
import numpy as np
import pytest
import torch

import oracle
from gpu_utils import DEV, lego_like, n, scene, t

pytestmark = pytest.mark.gpu


def test_ray_aabb_intersect_bit_exact_and_vs_twin(golden):
    from nerfacc_amd.grid import _ray_aabb_intersect, ray_aabb_intersect

    o, d, boxes = golden["k1_rays_o"], golden["k1_rays_d"], golden["k1_aabbs"]
    tmin, tmax, hits = ray_aabb_intersect(t(o), t(d), t(boxes))
    r_tmin, r_tmax, r_hits = oracle.ray_aabb_intersect(o, d, boxes)
    assert np.array_equal(n(hits), r_hits)
    assert np.array_equal(n(tmin), r_tmin) and np.array_equal(n(tmax), r_tmax)
    # reference: tests/test_grid.py:23-27 (kernel vs pure-torch twin)
    a, b, c = _ray_aabb_intersect(t(o), t(d), t(boxes))
    assert torch.allclose(tmin, a) and torch.allclose(tmax, b) and (hits == c).all()
    np.testing.assert_allclose(n(tmin), golden["k1_tmin"], rtol=1e-5)
    # near/far/miss arguments
    tmin, tmax, hits = ray_aabb_intersect(t(o), t(d), t(boxes), 0.1, 0.7, -1.0)
    r = oracle.ray_aabb_intersect(o, d, boxes, 0.1, 0.7, -1.0)
    assert np.array_equal(n(tmin), r[0]) and np.array_equal(n(tmax), r[1]) and np.array_equal(n(hits), r[2])


def _compare(iv, sm, term, r_iv, r_sm, r_term):
    assert np.array_equal(n(sm.packed_info), r_sm["packed_info"])
    assert np.array_equal(n(iv.packed_info), r_iv["packed_info"])
    assert np.array_equal(n(sm.ray_indices), r_sm["ray_indices"])
    assert np.array_equal(n(iv.ray_indices), r_iv["ray_indices"])
    assert np.array_equal(n(iv.is_left), r_iv["is_left"]) and np.array_equal(n(iv.is_right), r_iv["is_right"])
    assert np.array_equal(n(iv.vals), r_iv["vals"])          # float, bit-exact
    assert np.array_equal(n(sm.vals), r_sm["vals"])
    assert np.array_equal(n(sm.is_valid), r_sm["is_valid"])
    # rays without samples are skipped by the fill pass (grid.cu:103-106), so their
    # terminate_planes entry is never written — in the reference as well: compare the rest
    live = r_sm["packed_info"][:, 1] > 0
    assert np.array_equal(n(term)[live], r_term[live])


@pytest.mark.parametrize("kw", [
    dict(),                                               # tests/test_grid.py:38-68 configuration
    dict(step_size=0.02, cone_angle=0.004),
    dict(step_size=-1.0),                                 # one interval per occupied voxel
    dict(step_size=5e-3, near=0.3, far=2.5),
    dict(step_size=1e-2, traverse_steps_limit=37),
])
@pytest.mark.parametrize("levels,res", [(4, 32), (1, 64), (2, (20, 33, 7))])
def test_traverse_grids_bit_exact(kw, levels, res):
    from nerfacc_amd.grid import traverse_grids

    kw = dict(kw)
    res3 = (res,) * 3 if isinstance(res, int) else res
    o, d, aabbs, _ = scene(42 + levels, n_rays=300, levels=levels, res=8)
    binaries = np.random.default_rng(levels).random((levels,) + res3) < 0.4
    near, far = kw.pop("near", None), kw.pop("far", None)
    extra = {}
    if near is not None:
        rng = np.random.default_rng(0)
        extra["near_planes"] = (near * rng.random(300)).astype(np.float32)
        extra["far_planes"] = (far * (0.5 + rng.random(300))).astype(np.float32)
    r_iv, r_sm, r_term = oracle.traverse_grids(o, d, binaries, aabbs, **extra, **kw)
    g_extra = {k: t(v) for k, v in extra.items()}
    iv, sm, term = traverse_grids(t(o), t(d), t(binaries), t(aabbs), **g_extra, **kw)
    assert r_sm["vals"].shape[0] > 50
    _compare(iv, sm, term, r_iv, r_sm, r_term)
    # same result when the caller supplies the sorted intersections (grid.py:156-162)
    from nerfacc_amd.grid import ray_aabb_intersect

    tmin, tmax, hits = ray_aabb_intersect(t(o), t(d), t(aabbs))
    ts, ti = torch.sort(torch.cat([tmin, tmax], -1), dim=-1, stable=True)
    iv2, sm2, term2 = traverse_grids(t(o), t(d), t(binaries), t(aabbs), **g_extra, **kw, t_sorted=ts, t_indices=ti, hits=hits)
    _compare(iv2, sm2, term2, r_iv, r_sm, r_term)


def test_traverse_lego_like_128_bit_exact_and_invariants():
    from nerfacc_amd.grid import _query, traverse_grids

    o, d, aabb, occ = lego_like(0, 4096)
    r_iv, r_sm, r_term = oracle.traverse_grids(o, d, occ, aabb, step_size=5e-3)
    iv, sm, term = traverse_grids(t(o), t(d), t(occ), t(aabb), step_size=5e-3)
    assert r_sm["vals"].shape[0] > 100000
    _compare(iv, sm, term, r_iv, r_sm, r_term)
    # reference property (tests/test_grid.py:57-68): every sample midpoint is in an occupied cell
    ts_, te_ = iv.vals[iv.is_left], iv.vals[iv.is_right]
    pos = t(o)[sm.ray_indices] + t(d)[sm.ray_indices] * ((ts_ + te_)[:, None] / 2.0)
    occs, sel = _query(pos, t(occ), t(aabb[0]))
    assert sel.all() and occs.float().mean() > 0.9999


def test_traverse_test_mode_over_allocate():
    # reference: tests/test_grid.py:71-131
    from nerfacc_amd.grid import traverse_grids
    from nerfacc_amd.volrend import accumulate_along_rays

    o, d, aabbs, binaries = scene(42, n_rays=10)
    O, D, A, B = t(o), t(d), t(aabbs), t(binaries)
    iv, sm, _ = traverse_grids(O, D, B, A)
    acc_s = accumulate_along_rays(iv.vals[iv.is_left], None, sm.ray_indices, 10)
    acc_e = accumulate_along_rays(iv.vals[iv.is_right], None, sm.ray_indices, 10)
    a_s, a_e, near, mask = 0.0, 0.0, None, None
    r_near, r_mask = None, None
    for it in range(2):
        iv2, sm2, near = traverse_grids(O, D, B, A, near_planes=near, traverse_steps_limit=4000, over_allocate=True,
                                        rays_mask=mask)
        r_iv2, r_sm2, r_near = oracle.traverse_grids(o, d, binaries, aabbs, near_planes=r_near, traverse_steps_limit=4000,
                                                     over_allocate=True, rays_mask=r_mask)
        assert np.array_equal(n(sm2.packed_info), r_sm2["packed_info"])
        assert np.array_equal(n(iv2.vals), r_iv2["vals"]) and np.array_equal(n(sm2.is_valid), r_sm2["is_valid"])
        assert np.array_equal(n(iv2.is_left), r_iv2["is_left"]) and np.array_equal(n(iv2.is_right), r_iv2["is_right"])
        mask = sm2.packed_info[:, 1] == 4000
        r_mask = r_sm2["packed_info"][:, 1] == 4000
        if it == 0:
            assert np.array_equal(n(near), r_near)
        ri2 = sm2.ray_indices[sm2.is_valid]
        a_s = a_s + accumulate_along_rays(iv2.vals[iv2.is_left], None, ri2, 10)
        a_e = a_e + accumulate_along_rays(iv2.vals[iv2.is_right], None, ri2, 10)
    assert (~mask).all()
    assert torch.allclose(a_s, acc_s, atol=1e-1) and torch.allclose(a_e, acc_e, atol=1e-1)


def test_traverse_near_far_single_cell():
    # reference: tests/test_grid.py:134-160
    from nerfacc_amd.grid import traverse_grids

    o = torch.tensor([[-1.0, 0.0, 0.0]], device=DEV)
    d = torch.tensor([[1.0, 0.01, 0.01]], device=DEV)
    d = d / d.norm(dim=-1, keepdim=True)
    iv, sm, _ = traverse_grids(o, d, torch.ones((1, 1, 1, 1), dtype=torch.bool, device=DEV),
                               torch.tensor([[0.0, 0.0, 0.0, 1.0, 1.0, 1.0]], device=DEV),
                               step_size=0.05, near_planes=torch.tensor([1.2], device=DEV),
                               far_planes=torch.tensor([1.5], device=DEV))
    assert iv.vals.numel() > 0
    assert (iv.vals >= 1.2 - 0.025).all() and (iv.vals <= 1.5 + 0.025).all()


def test_traverse_empty_inputs_and_errors():
    from nerfacc_amd.grid import traverse_grids

    B = torch.zeros((1, 8, 8, 8), dtype=torch.bool, device=DEV)
    A = torch.tensor([[0.0, 0, 0, 1, 1, 1]], device=DEV)
    iv, sm, term = traverse_grids(torch.rand(5, 3, device=DEV), torch.ones(5, 3, device=DEV) / 3**0.5, B, A)
    assert iv.vals.numel() == 0 and sm.vals.numel() == 0 and (sm.packed_info == 0).all()
    iv, sm, term = traverse_grids(torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV), B, A)
    assert sm.vals.numel() == 0 and sm.packed_info.shape == (0, 2)
    with pytest.raises(RuntimeError):   # CPU tensors are rejected, there is no fallback
        traverse_grids(torch.rand(5, 3), torch.rand(5, 3), B.cpu(), A.cpu())
    with pytest.raises(AssertionError):
        traverse_grids(torch.rand(5, 3, device=DEV), torch.rand(5, 3, device=DEV), B, A, over_allocate=True)


def test_traverse_large_grid_global_occupancy_path():
    """2 levels of 256^3 (BASELINE configs[4] resolution): the brick bitmap no longer fits the
    LDS budget, so the kernels read the sparse occupancy from L2 instead — same results."""
    from nerfacc_amd.grid import traverse_grids

    rng = np.random.default_rng(7)
    o, d, aabbs, _ = scene(11, n_rays=200, levels=2, res=8)
    g = (np.arange(256) + 0.5) / 256 * 2 - 1
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    shell = (np.abs(np.sqrt(X**2 + Y**2 + Z**2) - 0.6) < 0.05) | ((np.abs(X) < 0.02) & (np.abs(Y) < 0.5))
    binaries = np.stack([shell, rng.random((256, 256, 256)) < 0.02])
    for kw in (dict(step_size=4e-3), dict(step_size=4e-3, cone_angle=0.003)):
        r_iv, r_sm, r_term = oracle.traverse_grids(o, d, binaries, aabbs, **kw)
        iv, sm, term = traverse_grids(t(o), t(d), t(binaries), t(aabbs), **kw)
        assert r_sm["vals"].shape[0] > 1000
        _compare(iv, sm, term, r_iv, r_sm, r_term)


def test_sampling_many_rays_serial_and_split_walks_agree():
    """the count pass uses 16 lanes per ray below ~20 k rays and one lane per ray above: both
    must give the oracle's answer (same scene, 4 k and 30 k rays)"""
    from nerfacc_amd import cuda as C

    for n_rays in (4000, 30000):
        o, d, aabb, occ = lego_like(5, n_rays)
        near = np.zeros(n_rays, np.float32)
        far = np.full(n_rays, 1e10, np.float32)
        ri, ts, te, pk = C.sample_occgrid(t(o), t(d), t(occ), t(aabb), t(near), t(far), 5e-3, 0.0)
        r_ri, r_ts, r_te, r_pk = oracle.sampling(o, d, occ, aabb, render_step_size=5e-3)
        assert np.array_equal(n(ri), r_ri) and np.array_equal(n(ts), r_ts) and np.array_equal(n(te), r_te)
        assert np.array_equal(n(pk), r_pk)


@pytest.mark.parametrize("levels,res,kind", [(2, 32, "blob"), (4, 32, "blob"), (4, 16, "noise"), (5, 24, "noise"), (8, 8, "blob"), (3, (20, 33, 7), "noise")])
def test_sampling_multi_level_segment_and_serial_count_passes(force_options, levels, res, kind):
    """several levels, cone_angle = 0: the count pass with one lane per LEVEL SEGMENT of a ray (grid.hip:
    traverse_count_segments_kernel; 8 lanes per ray up to 4 levels, 16 up to 8) and the lane-per-ray one must both give the oracle's
    samples, packed_info and terminate planes bit for bit — rays from inside the first level, from outside everything and
    axis-aligned; a noise grid overflows the segments' boundary lists (the ray then takes the serial walk inside the kernel)
    and the per-ray run records (those rays are re-walked by the fill pass)"""
    from nerfacc_amd import cuda as C

    rng = np.random.default_rng(levels * 100 + (res if isinstance(res, int) else sum(res)))
    res3 = (res,) * 3 if isinstance(res, int) else res
    c = [(np.arange(r) + 0.5) / r * 2 - 1 for r in res3]
    X, Y, Z = np.meshgrid(*c, indexing="ij")
    if kind == "blob":
        occ = np.stack([((X * 2.0**l) ** 2 + (Y * 2.0**l) ** 2 + (Z * 2.0**l) ** 2 < 0.5**2) | (rng.random(res3) < 0.004 * (l > 0)) for l in range(levels)])
    else:
        occ = rng.random((levels,) + res3) < 0.35
    aabbs = np.stack([np.array([-1, -1, -1, 1, 1, 1], np.float32) * 2.0**l for l in range(levels)])
    R = 1500
    v = rng.standard_normal((R, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    o = np.where(np.arange(R)[:, None] % 3 == 0, 0.6 * v, np.where(np.arange(R)[:, None] % 3 == 1, v * 2.0 ** (levels + 1), v * 1.3))
    d = np.where(np.arange(R)[:, None] % 3 == 1, -v + 0.2 * rng.standard_normal((R, 3)), rng.standard_normal((R, 3)))
    d[::7, rng.integers(0, 3)] = 0.0
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o, d = o.astype(np.float32), d.astype(np.float32)
    near = (rng.random(R) * 0.2).astype(np.float32)
    far = np.where(rng.random(R) < 0.7, 1e10, 2.0 ** (levels - 1)).astype(np.float32)
    step = 7e-3
    r_iv, r_sm, r_term = oracle.traverse_grids(o, d, occ, aabbs, near, far, step, 0.0)
    r_ts, r_te = r_iv["vals"][r_iv["is_left"]], r_iv["vals"][r_iv["is_right"]]
    assert r_sm["ray_indices"].shape[0] > 2000
    live = r_sm["packed_info"][:, 1] > 0               # (rays without samples: the reference leaves their terminate plane unwritten)
    for seg in (1, 0):
        force_options(segments=seg)
        ri, ts, te, pk, term = C.sample_occgrid(t(o), t(d), t(occ), t(aabbs), t(near), t(far), step, 0.0, with_terminate_planes=True)
        assert np.array_equal(n(ri), r_sm["ray_indices"]) and np.array_equal(n(ts), r_ts) and np.array_equal(n(te), r_te)
        assert np.array_equal(n(pk), r_sm["packed_info"])
        assert np.array_equal(n(term)[live], r_term[live])


@pytest.mark.parametrize("n_", [1, 1000, 8192, 8193, 100000, 640000, 5_000_001])
def test_exclusive_sum_i64_all_sizes(n_):
    """data_spec.hpp:86-106 (chunk_starts = cumsum(cnts) - cnts, total): one workgroup up to 8192 counts, beyond that chunks
    scanned by a workgroup each with the chunk totals parked in place — the over-allocated test-time pass scans one count per ray
    of a frame"""
    import ctypes

    from nerfacc_amd.cuda import _backend

    L = _backend.load_library()
    g = torch.Generator(device=DEV).manual_seed(n_)
    cnts = torch.randint(0, 1 << 20, (n_,), device=DEV, generator=g, dtype=torch.int64)
    cnts[torch.rand(n_, device=DEV, generator=g) < 0.3] = 0
    starts = torch.full((n_,), -1, device=DEV, dtype=torch.int64)
    total = torch.full((1,), -1, device=DEV, dtype=torch.int64)
    s = torch.cuda.current_stream().cuda_stream
    for tot in (total, None):
        rc = L.nfa_exclusive_sum_i64(ctypes.c_void_p(cnts.data_ptr()), n_, ctypes.c_void_p(starts.data_ptr()),
                                     ctypes.c_void_p(tot.data_ptr()) if tot is not None else None, ctypes.c_void_p(s))
        assert rc == 0
        want = torch.cumsum(cnts, 0) - cnts
        assert torch.equal(starts, want)
    assert int(total) == int(cnts.sum())


def test_sample_positions_bit_identical_to_the_torch_expression():
    import nerfacc_amd as nerfacc

    torch.manual_seed(3)
    R, N = 777, 50000
    o = torch.randn(R, 3, device="cuda:0")
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda:0"), dim=-1)
    ri = torch.sort(torch.randint(0, R, (N,), device="cuda:0"))[0]
    ts = torch.rand(N, device="cuda:0") * 5
    te = ts + torch.rand(N, device="cuda:0") * 0.01
    want_d = d[ri]
    want = o[ri] + want_d * ((ts + te)[:, None] / 2.0)
    pos, dirs = nerfacc.sample_positions(o, d, ri, ts, te, return_dirs=True)
    assert torch.equal(pos, want) and torch.equal(dirs, want_d)
    assert torch.equal(nerfacc.sample_positions(o, d, ri, ts, te), want)
    # rays that require grad go through the torch expression (autograd sees it)
    o2 = o.clone().requires_grad_(True)
    p2 = nerfacc.sample_positions(o2, d, ri, ts, te)
    p2.sum().backward()
    assert torch.equal(p2.detach(), want) and o2.grad is not None and float(o2.grad.sum()) == float(N * 3)
    assert nerfacc.sample_positions(o, d, ri[:0], ts[:0], te[:0]).shape == (0, 3)


@pytest.mark.parametrize("rb", [None, 2, 4, 5, 6])
def test_tile_emit_sub_blocks_and_rounds_on_a_noise_grid(force_options, rb):
    """the tile form of the emit pass (emit_pass.hpp) where a wave's block of rays has more runs than its segment list holds —
    sub-blocks of whole rays — and a ray has more runs than lanes share it — several rounds, the next round's records requested a
    round ahead: 70 000 rays through a 64^3 noise grid (30-60 runs per ray, 2 000-4 000 per 64-ray block against a list of 512),
    every rays-per-wave setting, against the lane-per-sample form (which the reference fixtures and the fuzz pin)"""
    from nerfacc_amd import cuda as C

    g = np.random.default_rng(21)
    occ = t(g.random((1, 64, 64, 64)) > 0.5)
    aabb = t(np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32))
    R = 70000
    o = g.standard_normal((R, 3)); o = (4.0 * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    d = (g.random((R, 3)) * 2.4 - 1.2) - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o, d = t(o), t(d)
    near, far = torch.zeros(R, device=DEV), torch.full((R,), 1e10, device=DEV)
    mask = t(g.random(R) < 0.6)
    calls = {"training": lambda: C.sample_occgrid(o, d, occ, aabb, near, far, 5e-3, 0.0),
             "marcher round": lambda: C.sample_occgrid(o, d, occ, aabb, near, far, 5e-3, 0.0, rays_mask=mask, traverse_steps_limit=24)}
    for name, call in calls.items():
        force_options(emit="samples", emit_rb=None)
        want = call()
        assert want[0].shape[0] > 500000
        force_options(emit="tiles", emit_rb=rb)
        for _ in range(2):                       # (the second call takes the speculative launch)
            got = call()
            assert all(torch.equal(a_, b_) for a_, b_ in zip(want, got)), (name, rb)
    # the reference-API call with interval outputs (edges, flags: the tile form's per-element instance) on the first 20 000 rays
    from nerfacc_amd.grid import traverse_grids

    def api():
        iv, sm, term = traverse_grids(o[:20000], d[:20000], occ, aabb, step_size=5e-3)
        return (iv.vals, iv.ray_indices, iv.is_left, iv.is_right, iv.packed_info, sm.vals, sm.ray_indices, sm.packed_info)

    force_options(emit="samples", emit_rb=None)
    want = api()
    force_options(emit="tiles", emit_rb=rb)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(want, api())), ("traverse_grids", rb)
