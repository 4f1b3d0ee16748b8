"""Occupancy-grid maintenance kernels (csrc/occgrid.hip) vs the oracle restatement of
OccGridEstimator._update (occ_grid.py:366-404), through the C ABI (nerfacc_amd.cuda)."""
import numpy as np
import pytest
import torch

import oracle
from gpu_utils import n, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("res", [(16, 16, 16), (32, 8, 20)])
def test_cell_points_bit_exact(res):
    from nerfacc_amd import cuda as C

    rng = np.random.default_rng(0)
    cells = res[0] * res[1] * res[2]
    aabb = np.array([-1.5, -0.5, 0.25, 1.0, 2.5, 3.0], np.float32)
    for ids in (None, rng.integers(0, cells, 5000), np.arange(cells)[::-1].copy()):
        m = cells if ids is None else len(ids)
        jit = rng.random((m, 3), dtype=np.float32)
        want = oracle.grid_cell_points(ids, jit, res, aabb)
        got = C.grid_cell_points(None if ids is None else t(ids.astype(np.int64)), t(jit), res, t(aabb))
        assert np.array_equal(n(got), want)          # float32, same operation order: bit-exact
    with pytest.raises(RuntimeError):
        C.grid_cell_points(None, torch.rand(4, 3), res, t(aabb))          # host tensor: no CPU path


def test_ema_update_and_repeated_ids():
    from nerfacc_amd import cuda as C

    rng = np.random.default_rng(1)
    cells = 4096
    occs = rng.random(cells, dtype=np.float32)
    occs[rng.integers(0, cells, 300)] = -1.0                       # invisible cells keep -1 * decay vs occ
    # (a) a permutation: no repeats, bit-exact
    ids = rng.permutation(cells)[:3000].astype(np.int64)
    occ_new = (rng.random(3000, dtype=np.float32) * 1.2).astype(np.float32)
    want = oracle.grid_ema_update(occs, ids, occ_new, 0.95)
    g = t(occs.copy())
    C.grid_ema_update(g, t(ids), t(occ_new), 0.95)
    assert np.array_equal(n(g), want)
    # (b) all cells (ids = None)
    occ_all = rng.random(cells, dtype=np.float32)
    g = t(occs.copy())
    C.grid_ema_update(g, None, t(occ_all), 0.5)
    assert np.array_equal(n(g), oracle.grid_ema_update(occs, None, occ_all, 0.5))
    # (c) repeats (uniform draws with replacement, occ_grid.py:350-352): every written cell holds one
    # of the candidates formed from the OLD grid; cells not named are untouched
    ids = rng.integers(0, 512, 6000).astype(np.int64)
    occ_new = rng.random(6000, dtype=np.float32)
    cand = oracle.grid_ema_candidates(occs, ids, occ_new, 0.95)
    g = t(occs.copy())
    C.grid_ema_update(g, t(ids), t(occ_new), 0.95)
    got = n(g)
    assert np.array_equal(got[512:], occs[512:])
    for c in range(512):
        sel = ids == c
        if sel.any():
            assert got[c] in cand[sel]
        else:
            assert got[c] == occs[c]


def test_threshold_matches_reference_rule():
    from nerfacc_amd import cuda as C

    rng = np.random.default_rng(2)
    occs = (rng.random(2 * 32**3, dtype=np.float32) ** 4 * 0.05).astype(np.float32)
    occs[rng.integers(0, len(occs), 5000)] = -1.0
    for occ_thre in (0.01, 1e-4, 10.0):
        want_bin, want_thre = oracle.grid_threshold(occs, occ_thre)
        got_bin, got_thre = C.grid_threshold(t(occs), occ_thre)
        assert got_bin.dtype == torch.bool
        thre = float(got_thre.item())
        assert abs(thre - float(want_thre)) <= 1e-6 * abs(float(want_thre))
        near = np.abs(occs - want_thre) <= 1e-6 * abs(float(want_thre))      # cells within rounding of the threshold
        assert np.array_equal(n(got_bin)[~near], want_bin[~near])
        # and the torch composition the reference runs on the device
        o = t(occs)
        ref = o > torch.clamp(o[o >= 0].mean(), max=occ_thre)
        assert (ref.cpu().numpy()[~near] == n(got_bin)[~near]).all()
    allneg = np.full(1000, -1.0, np.float32)                        # nothing visible: NaN mean, nothing passes
    b, th = C.grid_threshold(t(allneg), 0.01)
    assert not b.any() and torch.isnan(th).all()


@pytest.mark.parametrize("shape", [(2, 32, 32, 32), (1, 30, 31, 33), (3, 8, 12, 20)])
def test_threshold_packed_equals_threshold_then_pack(shape):
    """grid_threshold(shape=...) — threshold and bit-pack in one pass, cache entry included — gives the bool grid of the
    two-step form and, word for word, the packed grid nfa_pack_binaries builds from it (bricks, header with the per-level
    voxel counts, coarse bitmap, rank prefix, compacted bricks)"""
    from nerfacc_amd import cuda as C
    from nerfacc_amd.cuda import _backend

    rng = np.random.default_rng(sum(shape))
    ncell = int(np.prod(shape))
    occs = (rng.random(ncell, dtype=np.float32) ** 4 * 0.05).astype(np.float32)
    occs[rng.integers(0, ncell, ncell // 20)] = -1.0
    o = t(occs)
    flat, thre_a = C.grid_threshold(o, 0.01)
    fused, thre_b = C.grid_threshold(o, 0.01, shape)
    assert fused.shape == tuple(shape) and fused.dtype == torch.bool
    assert torch.equal(fused.reshape(-1), flat) and torch.equal(thre_a, thre_b)
    packed_fused = _backend.packed_bricks(fused).clone()            # cache hit: what the fused pass wrote
    packed_ref = _backend.packed_bricks(flat.view(shape).clone())   # a different tensor: packed from the bool grid
    assert packed_fused.data_ptr() != packed_ref.data_ptr()
    # the written parts of the buffer (grid.hip, PackedLayout); padding words are left as allocated
    G, rx, ry, rz = shape
    nb = G * ((rx + 3) // 4) * ((ry + 3) // 4) * ((rz + 3) // 4)
    nw = (nb + 31) // 32
    off_coarse = nb + 12
    off_prefix = off_coarse + (nw + 1) // 2
    off_compact = off_prefix + (nw + 1) // 2
    nonempty = int(packed_ref[nb])
    assert nonempty == int((packed_ref[:nb] != 0).sum())
    for a, b in ((0, nb + 9), (off_compact, off_compact + nonempty)):
        assert torch.equal(packed_fused[a:b], packed_ref[a:b])
    f32, r32 = packed_fused.view(torch.int32), packed_ref.view(torch.int32)
    for off in (off_coarse, off_prefix):
        assert torch.equal(f32[2 * off:2 * off + nw], r32[2 * off:2 * off + nw])
    assert C.grid_occupied_counts(fused) == [int(x) for x in fused.reshape(shape[0], -1).sum(1).tolist()]
    allneg = torch.full((int(np.prod(shape)),), -1.0, device=o.device)
    b, th = C.grid_threshold(allneg, 0.01, shape)
    assert not b.any() and torch.isnan(th).all()


def test_update_end_to_end_vs_oracle():
    """OccGridEstimator._update on the device == the oracle replay fed with the same draws of the
    device generator (same calls, same order: occ_grid.py:345-404)."""
    from nerfacc_amd import OccGridEstimator

    res, levels = 16, 2
    cells = res**3
    est = OccGridEstimator(roi_aabb=[-1.0, -1, -1, 1, 1, 1], resolution=res, levels=levels).to(DEV)
    occ_fn = lambda x: torch.exp(-2.0 * (x**2).sum(-1, keepdim=True)) * 0.05
    occs = np.zeros(levels * cells, np.float32)
    binaries = np.zeros(levels * cells, bool)
    aabbs = n(est.aabbs)
    for step in (0, 16, 256, 272):
        torch.manual_seed(100 + step)
        est._update(step=step, occ_eval_fn=occ_fn, occ_thre=0.01)
        torch.manual_seed(100 + step)
        lvl_ids = []
        if step < 256:
            lvl_ids = [np.nonzero(occs[l * cells:(l + 1) * cells] >= 0)[0] for l in range(levels)]
        else:
            q = cells // 4
            for l in range(levels):
                uni = n(torch.randint(cells, (q,), device=DEV))
                uni = uni[occs[l * cells + uni] >= 0]
                occd = np.nonzero(binaries[l * cells:(l + 1) * cells])[0]
                if q < len(occd):
                    occd = occd[n(torch.randint(len(occd), (q,), device=DEV))]
                lvl_ids.append(np.concatenate([uni, occd]))
        for l, ids in enumerate(lvl_ids):
            jitter = n(torch.rand((len(ids), 3), device=DEV))
            pts = oracle.grid_cell_points(ids, jitter, (res, res, res), aabbs[l])
            occ = n(occ_fn(t(pts)).squeeze(-1))                      # the user's field stays a torch function
            lvl = occs[l * cells:(l + 1) * cells]
            new = oracle.grid_ema_update(lvl, ids, occ, 0.95)
            # repeated ids: accept whichever candidate the device kept
            got = n(est.occs[l * cells:(l + 1) * cells])
            uniq, cnt = np.unique(ids, return_counts=True)
            rep = np.zeros(cells, bool)
            rep[uniq[cnt > 1]] = True
            assert np.array_equal(got[~rep], new[~rep])
            cand = oracle.grid_ema_candidates(lvl, ids, occ, 0.95)
            for c in uniq[cnt > 1]:
                assert got[c] in cand[ids == c]
            occs[l * cells:(l + 1) * cells] = got
        binaries, thre = oracle.grid_threshold(occs, 0.01)
        near = np.abs(occs - thre) <= 1e-6 * abs(float(thre))
        assert np.array_equal(n(est.binaries).ravel()[~near], binaries[~near])
        binaries = n(est.binaries).ravel().copy()
    assert 0 < binaries.sum() < binaries.size


def test_mark_invisible_cells_device_kernel_vs_reference_composition(golden):
    """device kernel (occgrid.hip) vs the reference's torch composition on host tensors (occ_grid.py:262-332, pinned bit for
    bit by tests/golden: 77660 / 53412 cells of tests/test_grid.py:232-233)"""
    from nerfacc_amd import OccGridEstimator

    base = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    K = torch.tensor([[[100.0, 0, 50.0], [0, 100.0, 50.0], [0, 0, 1]]])
    pose = torch.tensor([[[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5]]])
    est = OccGridEstimator(roi_aabb=base, resolution=32, levels=4).to(DEV)
    est.mark_invisible_cells(K.to(DEV), pose.to(DEV), 100, 100)
    occs = est.occs.cpu()
    assert (occs == -1).sum() == 77660 and (occs == 0).sum() == 53412
    assert np.array_equal(np.packbits((occs == -1).numpy()), golden["mic_occs_bits"])
    # many cameras, per-camera intrinsics, a near plane, a non-cubic grid: device vs host composition
    g = torch.Generator().manual_seed(5)
    C = 37
    pos = torch.randn(C, 3, generator=g)
    pos = 3.0 * pos / pos.norm(dim=-1, keepdim=True)
    fwd = -pos / pos.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm(dim=-1, keepdim=True)
    tup = torch.linalg.cross(right, fwd)
    c2w = torch.cat([torch.stack([right, -tup, fwd], -1), pos[:, :, None]], -1)        # OpenCV camera: +z forward
    Ks = torch.tensor([[60.0, 0, 40.0], [0, 60.0, 30.0], [0, 0, 1]]).repeat(C, 1, 1) * (1 + 0.1 * torch.rand(C, 1, 1, generator=g))
    Ks[:, 2, 2] = 1.0
    host = OccGridEstimator(roi_aabb=[-1.0, -0.8, -0.6, 1.0, 0.8, 0.6], resolution=[24, 20, 16], levels=2)
    host.mark_invisible_cells(Ks, c2w, 80, 60, near_plane=1.4)
    dev = OccGridEstimator(roi_aabb=[-1.0, -0.8, -0.6, 1.0, 0.8, 0.6], resolution=[24, 20, 16], levels=2).to(DEV)
    dev.mark_invisible_cells(Ks.to(DEV), c2w.to(DEV), 80, 60, near_plane=1.4)
    a, b = host.occs, dev.occs.cpu()
    assert 0.05 < (a == -1).float().mean() < 0.95
    assert (a != b).sum().item() <= 2, (a != b).sum().item()      # only a projection within an ulp of an image border may flip
    # cells marked invisible are excluded from the next update
    dev._update(step=0, occ_eval_fn=lambda x: torch.full((x.shape[0], 1), 0.5, device=DEV))
    assert (dev.occs[b.to(DEV) == -1] == -1).all() and (dev.occs[b.to(DEV) == 0] > 0).all()


def test_update_cell_selection_without_nonzero_sync_matches_reference_composition():
    """after warm-up `_update` draws n uniform cells + (at most n of) the occupied ones (occ_grid.py:345-364); the device path
    sizes `nonzero` from the packed grid's header instead of syncing — same cells, same RNG stream as the torch composition"""
    from nerfacc_amd import OccGridEstimator
    from nerfacc_amd.cuda import grid_occupied_counts

    torch.manual_seed(0)
    est = OccGridEstimator(roi_aabb=[-1.0, -1, -1, 1, 1, 1], resolution=32, levels=3).to(DEV)
    est.binaries = (torch.rand(3, 32, 32, 32, device=DEV) < torch.tensor([0.02, 0.3, 0.0], device=DEV)[:, None, None, None])
    cnts = grid_occupied_counts(est.binaries)
    assert cnts == [int(est.binaries[l].sum().item()) for l in range(3)] and cnts[2] == 0
    n = est.cells_per_lvl // 4
    torch.manual_seed(11)
    got = est._sample_uniform_and_occupied_cells(n)
    torch.manual_seed(11)
    want = []
    for lvl in range(3):                                  # occ_grid.py:345-364 as written there
        uniform = torch.randint(est.cells_per_lvl, (n,), device=DEV)
        uniform = uniform[est.occs[lvl * est.cells_per_lvl + uniform] >= 0.0]
        occupied = torch.nonzero(est.binaries[lvl].flatten())[:, 0]
        if n < len(occupied):
            occupied = occupied[torch.randint(len(occupied), (n,), device=DEV)]
        want.append(torch.cat([uniform, occupied], dim=0))
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,p", [((1, 128, 128, 128), 0.06), ((2, 20, 28, 12), 0.5), ((1, 5, 7, 3), 0.9), ((3, 64, 64, 64), 0.001),
                                     ((1, 256, 256, 256), 0.02), ((4, 32, 32, 32), 1.0), ((2, 16, 16, 16), 0.0)])
def test_occupied_cells_equal_nonzero(shape, p):
    """round 6: the list of a level's occupied cells (OccGridEstimator._sample_uniform_and_occupied_cells, occ_grid.py:356) is one
    launch of the library — ranks by popcount + a look-back over the workgroups, C ABI nfa_grid_occupied_cells — instead of
    torch.nonzero_static: exactly `torch.nonzero(binaries[lvl].flatten())[:, 0]`, level after level, twenty times in a row (the sync
    block is shared with the sampling call), in both host faces"""
    from nerfacc_amd import cuda as C
    from nerfacc_amd.cuda import _backend

    g = torch.Generator().manual_seed(sum(shape))
    binaries = (torch.rand(shape, generator=g) < p).to(DEV)
    want = [torch.nonzero(binaries[l].flatten())[:, 0] for l in range(shape[0])]
    for rep in range(20 if shape[1] <= 64 else 3):
        for l in range(shape[0]):
            got = C.grid_occupied_cells(binaries, l)
            assert got.dtype == torch.int64 and torch.equal(got, want[l])
    for l in range(shape[0]):
        assert torch.equal(_backend._CtypesC.grid_occupied_cells(binaries, l), want[l])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 128, 128, 128), (2, 20, 28, 12), (1, 5, 7, 3), (3, 64, 64, 64)])
def test_brick_distance_field(shape):
    """round 5: the packed grid carries, per 4^3 brick, the Chebyshev distance (in bricks, within its level, capped at 4) to the nearest
    non-empty brick — what the empty-space macro steps of the count pass size their jumps from (grid.hip: brick_dist_kernel).
    Checked against scipy's chessboard distance transform of the brick bitmap."""
    from scipy import ndimage

    from nerfacc_amd.cuda import _backend

    rng = np.random.default_rng(3)
    G, rx, ry, rz = shape
    occ = np.zeros(shape, bool)
    for g in range(G):                                  # a few blobs and single voxels per level; level 1 of several stays empty
        if G > 1 and g == 1:
            continue
        for _ in range(4):
            c = rng.integers(0, [rx, ry, rz])
            r = rng.integers(0, 3)
            occ[g, max(c[0] - r, 0):c[0] + r + 1, max(c[1] - r, 0):c[1] + r + 1, max(c[2] - r, 0):c[2] + r + 1] = True
    packed = _backend.packed_bricks(t(occ)).cpu().numpy()
    nbx, nby, nbz = (rx + 3) // 4, (ry + 3) // 4, (rz + 3) // 4
    nb = G * nbx * nby * nbz
    nw = (nb + 31) // 32
    off_dist = nb + 12 + 2 * ((nw + 1) // 2) + nb
    nib = packed[off_dist:off_dist + (nb + 15) // 16].view(np.uint8)
    dist = np.stack([nib & 15, nib >> 4], -1).reshape(-1)[:nb].reshape(G, nbx, nby, nbz)
    pad = np.zeros((G, nbx * 4, nby * 4, nbz * 4), bool)
    pad[:, :rx, :ry, :rz] = occ
    nonempty = pad.reshape(G, nbx, 4, nby, 4, nbz, 4).any(axis=(2, 4, 6))
    for g in range(G):
        want = np.minimum(ndimage.distance_transform_cdt(~nonempty[g], metric="chessboard"), 4) if nonempty[g].any() else np.full((nbx, nby, nbz), 4)
        assert np.array_equal(dist[g], want), (shape, g)
